"""imdbn -- MI355X-native contrastive-divergence engine behind the reference's class API.

The package keeps the reference's dotted paths (``imdbn.models.rbm.RBM``,
``imdbn.models.gdbn_model_complete.iMDBN`` ...) so pickles move between the two
(SURVEY.md Appendix C).  Put ``multimodal-idbn_amd/`` on ``sys.path`` to import it.
"""
