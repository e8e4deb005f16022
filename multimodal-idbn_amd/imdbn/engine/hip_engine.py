"""HipEngine: torch tensors in, C-ABI calls out.  The only thing between ``RBM`` methods and the HIP kernels.

PyTorch is plumbing here (device memory, current stream); every arithmetic step of the hot path
runs in ``libimdbn_hip.so``.  Parameters are read from the RBM object at EVERY call (callers may
mutate or re-bind ``W``/biases behind the RBM's back); the engine holds only
scratch workspaces keyed on (device, V, H, B).
"""
from __future__ import annotations

import os
import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import native as N
from . import rng as R


def _f32c(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.size(1):
        return t
    if not t.is_contiguous():
        t = t.contiguous()
    return t


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


class HipEngine:
    name = "hip"
    fused_forward = True        # cd_step(forward=True): imdbn_cd_opts.fwd_out

    def __init__(self):
        self._lib = N.lib()
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._pf: Dict[tuple, tuple] = {}           # workspace key -> (identity of the prefetched batch, slot, tensor)
        self._pf_ok: Dict[tuple, bool] = {}
        self.mode = N.PARITY_F32
        # tuning / A-B aid: IMDBN_OPTS="name=value,..." -> imdbn_set_option (include/imdbn_engine.h)
        for kv in filter(None, os.environ.get("IMDBN_OPTS", "").split(",")):
            k, _, v = kv.partition("=")
            self.set_option(k.strip(), int(v or 1))

    # ---- plumbing -----------------------------------------------------------------------------
    def device_info(self):
        cu = C.c_int(0)
        buf = C.create_string_buffer(64)
        N.check(self._lib.imdbn_device_info(C.byref(cu), buf, 64), "imdbn_device_info")
        return cu.value, buf.value.decode()

    def set_tuning(self, ksplit_up: int = 0, ksplit_down: int = 0):
        N.check(self._lib.imdbn_set_tuning(int(ksplit_up), int(ksplit_down)), "imdbn_set_tuning")
        self._ws.clear(); self._pf.clear(); self._pf_ok.clear()

    def set_option(self, name: str, value: int):
        N.check(self._lib.imdbn_set_option(name.encode(), int(value)), "imdbn_set_option")
        self._ws.clear(); self._pf.clear(); self._pf_ok.clear()

    # per-caller knobs (imdbn_options): a handle bound to the calling thread overrides the process defaults
    def options_create(self, **knobs):
        h = C.c_void_p(self._lib.imdbn_options_create())
        if not h:
            raise N.EngineError("imdbn_options_create failed")
        for k, v in knobs.items():
            N.check(self._lib.imdbn_options_set(h, k.encode(), int(v)), "imdbn_options_set")
        return h

    def use_options(self, handle):
        """Bind `handle` (None: the process defaults) to this thread; workspaces are re-derived (the layout may differ)."""
        N.check(self._lib.imdbn_use_options(handle), "imdbn_use_options")
        self._ws.clear(); self._pf.clear(); self._pf_ok.clear()

    def options_destroy(self, handle):
        self._lib.imdbn_options_destroy(handle)

    def rng_advance(self, counter: torch.Tensor, n: int):
        """counter[0] += n on the current stream (a node of the graph being captured)."""
        assert counter.dtype == torch.int64 and counter.numel() == 1 and counter.is_cuda
        N.check(self._lib.imdbn_rng_advance(_ptr(counter), int(n), self._stream(counter.device)), "imdbn_rng_advance")

    def profile(self, on: bool):
        N.check(self._lib.imdbn_profile_enable(1 if on else 0), "imdbn_profile_enable")

    def profile_read(self) -> Tuple[float, int]:
        ms, n = C.c_double(0), C.c_int(0)
        N.check(self._lib.imdbn_profile_read(C.byref(ms), C.byref(n)), "imdbn_profile_read")
        return ms.value, n.value

    def debug_buffer(self, dev, V, H, B, name: str, nbytes: int) -> torch.Tensor:
        """Test aid: uint8 view of a named internal buffer of the (V, H, B) workspace (imdbn_debug_ws_offset)."""
        off = C.c_size_t(0)
        N.check(self._lib.imdbn_debug_ws_offset(int(V), int(H), int(B), name.encode(), C.byref(off)), "imdbn_debug_ws_offset")
        ws = self._workspace(torch.device(dev), V, H, B)
        return ws[off.value:off.value + nbytes]

    def _workspace(self, dev, V, H, B):
        # one workspace per (device, shape, STREAM): two same-shape RBMs driven from two streams must not share scratch
        key = (dev, V, H, B, torch.cuda.current_stream(dev).cuda_stream if torch.device(dev).type == "cuda" else 0)
        ws = self._ws.get(key)
        if ws is None:
            need = int(self._lib.imdbn_ws_bytes(V, H, B))
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._ws[key] = ws
        return ws

    @staticmethod
    def _stream(dev):
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def _desc(self, rbm, need_momentum: bool) -> N.RbmDesc:
        """The native descriptor of `rbm`.  Built (and validated) once per state of the parameter tensors: callers may re-bind
        or move W / biases / momentum buffers at any time (SURVEY b-1), so the cached descriptor is keyed on their addresses."""
        p = rbm._parameters
        Wp, hbp, vbp = p.get("W"), p.get("hid_bias"), p.get("vis_bias")
        dct = rbm.__dict__
        if Wp is not None and hbp is not None and vbp is not None:
            Wm, hbm, vbm = dct.get("W_m"), dct.get("hb_m"), dct.get("vb_m")
            g = dct.get("softmax_groups")
            key = (Wp.data_ptr(), Wp.stride(0), Wp.shape, hbp.data_ptr(), vbp.data_ptr(),
                   Wm.data_ptr() if isinstance(Wm, torch.Tensor) else 0, hbm.data_ptr() if isinstance(hbm, torch.Tensor) else 0,
                   vbm.data_ptr() if isinstance(vbm, torch.Tensor) else 0, Wm.stride(0) if isinstance(Wm, torch.Tensor) and Wm.dim() == 2 else 0,
                   tuple(map(tuple, g)) if g else (), self.mode, bool(need_momentum))
            cache = dct.get("_imdbn_desc")
            hit = cache.get(bool(need_momentum)) if cache is not None else None
            if hit is not None and hit[0] == key:
                return hit[1]
        d = self._build_desc(rbm, need_momentum)
        if Wp is not None and hbp is not None and vbp is not None:
            # (re-homing may have replaced the momentum buffers: key on what is there now)
            Wm, hbm, vbm = dct.get("W_m"), dct.get("hb_m"), dct.get("vb_m")
            g = dct.get("softmax_groups")
            key = (Wp.data_ptr(), Wp.stride(0), Wp.shape, hbp.data_ptr(), vbp.data_ptr(),
                   Wm.data_ptr() if isinstance(Wm, torch.Tensor) else 0, hbm.data_ptr() if isinstance(hbm, torch.Tensor) else 0,
                   vbm.data_ptr() if isinstance(vbm, torch.Tensor) else 0, Wm.stride(0) if isinstance(Wm, torch.Tensor) and Wm.dim() == 2 else 0,
                   tuple(map(tuple, g)) if g else (), self.mode, bool(need_momentum))
            dct.setdefault("_imdbn_desc", {})[bool(need_momentum)] = (key, d)
        return d

    def _build_desc(self, rbm, need_momentum: bool) -> N.RbmDesc:
        W = rbm.W.data
        if not W.is_cuda:
            raise N.EngineError("HipEngine needs CUDA/HIP tensors (RBM.W is on %s)" % W.device)
        if W.dtype != torch.float32 or W.dim() != 2 or W.stride(1) != 1:
            raise N.EngineError("RBM.W must be fp32 [V,H] with unit inner stride")
        V, H = W.shape
        d = N.RbmDesc()
        d.W, d.ldw, d.V, d.H = W.data_ptr(), W.stride(0), V, H
        hb, vb = rbm.hid_bias.data, rbm.vis_bias.data
        for t, n, nm in ((hb, H, "hid_bias"), (vb, V, "vis_bias")):
            if t.device != W.device or t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous():
                raise N.EngineError(f"RBM.{nm} must be contiguous fp32 [{n}] on {W.device}")
        d.hid_bias, d.vis_bias = hb.data_ptr(), vb.data_ptr()
        if need_momentum:
            # momentum buffers are plain attributes, not moved by .to() (rbm.py:77-79): re-home them
            for nm, ref in (("W_m", W), ("hb_m", hb), ("vb_m", vb)):
                m = getattr(rbm, nm, None)
                if m is None or m.shape != ref.shape:
                    m = torch.zeros_like(ref)
                elif m.device != ref.device or m.dtype != torch.float32:
                    m = m.to(device=ref.device, dtype=torch.float32)
                if nm != "W_m" and not m.is_contiguous():
                    m = m.contiguous()
                setattr(rbm, nm, m)
            if rbm.W_m.stride(0) != W.stride(0) or rbm.W_m.stride(1) != 1:
                # the C ABI has one leading dimension for W and W_m: re-home W_m with W's pitch (values kept)
                m2 = torch.empty_strided(tuple(W.shape), tuple(W.stride()), dtype=torch.float32, device=W.device)
                m2.copy_(rbm.W_m)
                rbm.W_m = m2
            d.W_m, d.hb_m, d.vb_m = rbm.W_m.data_ptr(), rbm.hb_m.data_ptr(), rbm.vb_m.data_ptr()
        groups = list(getattr(rbm, "softmax_groups", None) or [])
        if len(groups) > N.MAX_GROUPS:
            raise N.EngineError(f"at most {N.MAX_GROUPS} softmax groups supported")
        d.n_groups = len(groups)
        for i, (s, e) in enumerate(groups):
            d.group_start[i], d.group_end[i] = int(s), int(e)
        d.mode = self.mode
        return d

    @staticmethod
    def _groups(rbm):
        return [(int(s), int(e)) for s, e in (getattr(rbm, "softmax_groups", None) or [])]

    def _rng(self, rng, schedule: R.Schedule, B: int, dev):
        """native rng struct + keep-alive tensors"""
        r = N.Rng()
        keep = None
        if isinstance(rng, R.ReplayRng):
            ft, ct = rng.build(schedule, B, dev)
            r.mode = N.RNG_REPLAY
            r.tape, r.tape_len = ft.data_ptr(), ft.numel()
            r.cat_tape, r.cat_len = ct.data_ptr(), ct.numel()
            keep = (ft, ct)
        elif isinstance(rng, R.PhiloxRng):
            r.mode = N.RNG_PHILOX
            r.seed, r.offset, r.row0 = rng.seed, rng.offset, rng.row0
            if rng.device_counter is not None:          # a capture is being recorded: draw number = offset + *counter at run time
                r.dev_offset = rng.device_counter.data_ptr()
        else:
            raise N.EngineError("rng must be PhiloxRng or ReplayRng")
        return r, keep

    @staticmethod
    def _done(rng, r: N.Rng, schedule):
        if r.draws_used != len(schedule):
            raise N.EngineError(f"draw schedule mismatch: engine consumed {r.draws_used}, host planned {len(schedule)}")
        rng.advance(r.draws_used)

    def skip_draws(self, rng, schedule: R.Schedule, B: int):
        """Consume draws without computing (dead refinement passes of _cross_reconstruct)."""
        if isinstance(rng, R.ReplayRng):
            rng.build(schedule, B, "cpu")
        else:
            rng.advance(len(schedule))

    @staticmethod
    def _steps(steps: Sequence[dict]):
        arr = (N.ChainStep * max(1, len(steps)))()
        for i, s in enumerate(steps):
            arr[i].T, arr[i].sigma, arr[i].eta = float(s["T"]), float(s["sigma"]), float(s["eta"])
            arr[i].sample_h, arr[i].vmode, arr[i].clamp = int(s["sample_h"]), int(s["vmode"]), int(s["clamp"])
        return arr

    # ---- propagations -------------------------------------------------------------------------
    def prop_up(self, rbm, v, T=1.0, sample=False, rng=None):
        d = self._desc(rbm, False)
        v = _f32c(v, "v")
        B, dev = v.size(0), v.device
        out = torch.empty(B, d.H, device=dev)
        smp = torch.empty(B, d.H, device=dev) if sample else None
        sched = [("u", d.H)] if sample else []
        r, keep = self._rng(rng, sched, B, dev) if sample else (None, None)
        ws = self._workspace(dev, d.V, d.H, B)
        N.check(self._lib.imdbn_rbm_prop_up(C.byref(d), _ptr(v), v.stride(0), B, float(T), C.byref(r) if r else None,
                                             _ptr(out), out.stride(0), _ptr(smp), d.H, _ptr(ws), ws.numel(),
                                             self._stream(dev)), "imdbn_rbm_prop_up")
        if sample:
            self._done(rng, r, sched)
            return out, smp
        return out

    def forward(self, rbm, v, data_binary=None):
        """forward(v) at T = 1 (imdbn_rbm_forward): the streaming K1 reads 0/1 pieces of the batch as a bit plane."""
        d = self._desc(rbm, False)
        x = _f32c(v, "v")
        B, dev = x.size(0), x.device
        out = torch.empty(B, d.H, device=dev)
        ws = self._workspace(dev, d.V, d.H, B)
        binary = self._hint(v, data_binary)
        N.check(self._lib.imdbn_rbm_forward(C.byref(d), _ptr(x), x.stride(0), B, binary, _ptr(out), out.stride(0), _ptr(ws), ws.numel(),
                                             self._stream(dev)), "imdbn_rbm_forward")
        return out

    def free_energy(self, rbm, v):
        d = self._desc(rbm, False)
        v = _f32c(v, "v")
        B, dev = v.size(0), v.device
        out = torch.empty(B, device=dev)
        ws = self._workspace(dev, d.V, d.H, B)
        N.check(self._lib.imdbn_rbm_free_energy(C.byref(d), _ptr(v), v.stride(0), B, _ptr(out), _ptr(ws), ws.numel(),
                                                 self._stream(dev)), "imdbn_rbm_free_energy")
        return out

    def prop_down(self, rbm, h, T=1.0, logits_only=False):
        d = self._desc(rbm, False)
        h = _f32c(h, "h")
        B, dev = h.size(0), h.device
        out = torch.empty(B, d.V, device=dev)
        ws = self._workspace(dev, d.V, d.H, B)
        N.check(self._lib.imdbn_rbm_prop_down(C.byref(d), _ptr(h), h.stride(0), B, float(T), 1 if logits_only else 0,
                                               _ptr(out), out.stride(0), _ptr(ws), ws.numel(), self._stream(dev)),
                "imdbn_rbm_prop_down")
        return out

    def sample_visible(self, rbm, v_prob, rng):
        d = self._desc(rbm, False)
        p = _f32c(v_prob, "v_prob")
        B, dev = p.size(0), p.device
        out = torch.empty(B, d.V, device=dev)
        sched = R.sched_sample_visible(d.V, self._groups(rbm))
        r, keep = self._rng(rng, sched, B, dev)
        N.check(self._lib.imdbn_rbm_sample_visible(C.byref(d), _ptr(p), p.stride(0), B, C.byref(r), _ptr(out),
                                                    out.stride(0), self._stream(dev)), "imdbn_rbm_sample_visible")
        self._done(rng, r, sched)
        return out

    def gibbs_step(self, rbm, v, sample_h, sample_v, rng):
        d = self._desc(rbm, False)
        v = _f32c(v, "v")
        B, dev = v.size(0), v.device
        v_next, v_prob = torch.empty(B, d.V, device=dev), torch.empty(B, d.V, device=dev)
        h, h_prob = torch.empty(B, d.H, device=dev), torch.empty(B, d.H, device=dev)
        sched = ([("u", d.H)] if sample_h else []) + (R.sched_sample_visible(d.V, self._groups(rbm)) if sample_v else [])
        r, keep = self._rng(rng, sched, B, dev)
        ws = self._workspace(dev, d.V, d.H, B)
        N.check(self._lib.imdbn_rbm_gibbs_step(C.byref(d), _ptr(v), v.stride(0), B, int(bool(sample_h)), int(bool(sample_v)),
                                                C.byref(r), _ptr(v_next), _ptr(v_prob), _ptr(h), _ptr(h_prob),
                                                _ptr(ws), ws.numel(), self._stream(dev)), "imdbn_rbm_gibbs_step")
        self._done(rng, r, sched)
        return v_next, v_prob, h, h_prob

    # ---- CD updates ---------------------------------------------------------------------------
    @staticmethod
    def _opts(rbm, lr, mom, cd_k, sparsity=False, sample_h=False, sample_v=False, reclamp=True) -> N.CdOpts:
        o = N.CdOpts()
        o.cd_k, o.lr, o.momentum, o.weight_decay = int(cd_k), float(lr), float(mom), float(rbm.weight_decay)
        o.sparsity, o.sparsity_target = int(bool(sparsity)), float(getattr(rbm, "sparsity_factor", 0.0))
        o.sample_h, o.sample_v, o.reclamp_negative = int(bool(sample_h)), int(bool(sample_v)), int(bool(reclamp))
        return o

    @staticmethod
    def binary_hint(x: torch.Tensor) -> int:
        """What the HOST knows about the values of the batch `x` (imdbn_cd_opts.data_binary): ``N.DATA_BINARY`` / ``N.DATA_REAL``
        when the tensor carries the tag ``_imdbn_binary`` (``imdbn.datasets.DeviceLoader`` batches and sequential
        ``TensorDataset`` loaders: True; engine outputs -- probabilities -- : False), else ``N.DATA_UNKNOWN``.

        Unknown is the normal case (a training loop that builds a fresh tensor per step, idbn.py:199-203) and costs nothing:
        the device decides per 64-column piece of the batch, from the exactness map its own preparation writes, whether the
        positive phase reads it as a bit plane or as bf16 terms -- the same numbers either way, so results never depend on what
        the host knew or on what ran before.  The host never inspects a batch (no reduction, no synchronisation)."""
        tag = getattr(x, "_imdbn_binary", None)
        if tag is None:
            return N.DATA_UNKNOWN
        return N.DATA_BINARY if tag else N.DATA_REAL

    @staticmethod
    def _hint(x: torch.Tensor, given) -> int:
        """`given`: None (ask the tensor's tag), a bool (True: asserted 0/1; False: nothing known) or an N.DATA_* code."""
        if given is None:
            return HipEngine.binary_hint(x)
        if isinstance(given, bool):
            return N.DATA_BINARY if given else N.DATA_UNKNOWN
        return int(given)

    @staticmethod
    def _ident(t: torch.Tensor):
        """What must be unchanged for prefetched operand forms of `t` to be still valid (best effort: writes through
        ``.data`` or raw pointers do not bump the version -- the caller of ``next_data=`` promises not to do that)."""
        return (t.data_ptr(), t.untyped_storage().data_ptr(), t._version, tuple(t.shape), t.stride(0))

    def prefetch_ok(self, d, B) -> bool:
        key = (d.V, d.H, B, d.ldw, (d.W or 0) & 15, (d.W_m or 0) & 15)
        ok = self._pf_ok.get(key)
        if ok is None:
            ok = self._pf_ok[key] = bool(self._lib.imdbn_rbm_prefetch_ok(C.byref(d), B))
        return ok

    def _prefetch_opts(self, o, d, x, next_data):
        """Fill the next-batch fields of the options of a CD pass on batch `x` (imdbn_cd_opts.next_* / data_slot); returns the
        key of the prefetch state and the accepted hint (None: none) -- the caller records ``self._pf[key]`` after the call."""
        B, dev = x.size(0), x.device
        key = (dev, d.V, d.H, B, torch.cuda.current_stream(dev).cuda_stream)
        st = self._pf.pop(key, None)
        if st is not None and st[0] == self._ident(x):
            o.data_slot = st[1]
        nxt = None
        if (next_data is not None and next_data.dtype == torch.float32 and next_data.device == dev and next_data.dim() == 2
                and tuple(next_data.shape) == tuple(x.shape) and next_data.stride(1) == 1 and self.prefetch_ok(d, B)):
            nxt = next_data
            o.next_data, o.ld_next, o.next_slot = nxt.data_ptr(), nxt.stride(0), (2 if o.data_slot == 1 else 1)
            o.next_binary = self.binary_hint(next_data)
        return key, nxt

    def cd_step(self, rbm, data, lr, mom, cd_k, rng, next_data=None, data_binary=None, forward=False):
        """One CD-k update.  ``next_data``: the batch the NEXT cd_step of this shape will get -- its operand forms are
        then prepared by extra blocks of this call's first negative-phase launch and the
        next call skips its own preparation when it is handed that very tensor, unmodified.
        ``forward``: also return ``forward(data)`` under the updated weights (one more propagation in the same call)."""
        d = self._desc(rbm, True)
        x = _f32c(data, "data")
        B, dev = x.size(0), x.device
        o = self._opts(rbm, lr, mom, cd_k, sparsity=getattr(rbm, "sparsity", False))
        o.data_binary = self._hint(data, data_binary)
        sched = R.sched_cd(d.V, d.H, self._groups(rbm), cd_k)
        r, keep = self._rng(rng, sched, B, dev)
        loss = torch.empty(1, device=dev)
        ws = self._workspace(dev, d.V, d.H, B)
        key, nxt = self._prefetch_opts(o, d, x, next_data)
        fwd = None
        if forward:
            fwd = torch.empty(B, d.H, device=dev)
            o.fwd_out, o.ld_fwd = fwd.data_ptr(), fwd.stride(0)
        N.check(self._lib.imdbn_rbm_cd_step(C.byref(d), _ptr(x), x.stride(0), B, C.byref(o), C.byref(r), _ptr(loss),
                                             _ptr(ws), ws.numel(), self._stream(dev)), "imdbn_rbm_cd_step")
        self._done(rng, r, sched)
        if nxt is not None:                      # the strong reference keeps the address from being recycled
            self._pf[key] = (self._ident(nxt), int(o.next_slot), nxt)
        if forward:
            return loss.reshape(()), fwd
        return loss.reshape(())

    def assoc_update(self, rbm, vpos, hpos, vneg, hneg, lr, mom):
        """The weight / bias update alone (rbm.py:209-224) from the four phase tensors (imdbn_rbm_assoc_update)."""
        d = self._desc(rbm, True)
        ts = [_f32c(t, "t") for t in (vpos, hpos, vneg, hneg)]
        B, dev = ts[0].size(0), ts[0].device
        o = self._opts(rbm, lr, mom, 1, sparsity=getattr(rbm, "sparsity", False))
        ws = self._workspace(dev, d.V, d.H, B)
        N.check(self._lib.imdbn_rbm_assoc_update(C.byref(d), _ptr(ts[0]), ts[0].stride(0), _ptr(ts[1]), ts[1].stride(0), _ptr(ts[2]),
                                                  ts[2].stride(0), _ptr(ts[3]), ts[3].stride(0), B, C.byref(o), _ptr(ws), ws.numel(),
                                                  self._stream(dev)), "imdbn_rbm_assoc_update")

    # ---- RCCL through the C ABI (a binder without torch.distributed; the classes use torch.distributed) ---------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        N.check(self._lib.imdbn_comm_unique_id(buf), "imdbn_comm_unique_id")
        return buf.raw

    def comm_init(self, world: int, rank: int, uid: bytes):
        comm = C.c_void_p(0)
        N.check(self._lib.imdbn_comm_init(C.byref(comm), int(world), int(rank), C.create_string_buffer(uid, 128)), "imdbn_comm_init")
        return comm

    def comm_destroy(self, comm):
        N.check(self._lib.imdbn_comm_destroy(comm), "imdbn_comm_destroy")

    def comm_allreduce_sum(self, comm, t: torch.Tensor):
        assert t.dtype == torch.float32 and t.is_contiguous()
        N.check(self._lib.imdbn_allreduce_sum_f32(comm, _ptr(t), t.numel(), self._stream(t.device)), "imdbn_allreduce_sum_f32")
        return t

    def comm_allgather(self, comm, send: torch.Tensor, recv: torch.Tensor):
        assert send.is_contiguous() and recv.is_contiguous() and recv.numel() * recv.element_size() % (send.numel() * send.element_size()) == 0
        N.check(self._lib.imdbn_allgather_bytes(comm, _ptr(send), _ptr(recv), send.numel() * send.element_size(), self._stream(send.device)),
                "imdbn_allgather_bytes")
        return recv

    def packed_floats(self, V, H) -> int:
        return int(self._lib.imdbn_packed_delta_floats(V, H))

    def packed_buffer(self, rbm) -> torch.Tensor:
        """Reusable all-reduce buffer for this RBM's shape (every entry but the <=3 pad floats is rewritten by
        cd_stats, so no per-step zeroing)."""
        W = rbm.W.data
        key = ("packed", W.device, W.shape[0], W.shape[1])
        buf = self._ws.get(key)
        if buf is None:
            buf = self._ws[key] = torch.zeros(self.packed_floats(W.shape[0], W.shape[1]), device=W.device)
        return buf

    def cd_stats(self, rbm, data, cd_k, rng, out: Optional[torch.Tensor] = None, data_binary=None, next_data=None):
        """The shard's packed statistics (data-parallel all-reduce exchange); ``next_data``: the next-batch hint of ``cd_step``."""
        d = self._desc(rbm, True)       # (imdbn_rbm_prefetch_ok wants the full descriptor)
        x = _f32c(data, "data")
        B, dev = x.size(0), x.device
        o = self._opts(rbm, 0.0, 0.0, cd_k)
        o.data_binary = self._hint(data, data_binary)
        sched = R.sched_cd(d.V, d.H, self._groups(rbm), cd_k)
        r, keep = self._rng(rng, sched, B, dev)
        n = self.packed_floats(d.V, d.H)
        packed = out if out is not None else torch.zeros(n, device=dev)
        ws = self._workspace(dev, d.V, d.H, B)
        key, nxt = self._prefetch_opts(o, d, x, next_data)
        N.check(self._lib.imdbn_rbm_cd_stats(C.byref(d), _ptr(x), x.stride(0), B, C.byref(o), C.byref(r), _ptr(packed),
                                              _ptr(ws), ws.numel(), self._stream(dev)), "imdbn_rbm_cd_stats")
        self._done(rng, r, sched)
        if nxt is not None:
            self._pf[key] = (self._ident(nxt), int(o.next_slot), nxt)
        return packed

    # ---- data-parallel factor exchange (include/imdbn_engine.h) --------------------------------------
    def factor_block(self, V, H, B):
        """(offset, bytes) of the factor block inside the workspace of an (V, H, B) call."""
        off, nb = C.c_size_t(0), C.c_size_t(0)
        N.check(self._lib.imdbn_factor_block(int(V), int(H), int(B), C.byref(off), C.byref(nb)), "imdbn_factor_block")
        return int(off.value), int(nb.value)

    def factor_mode_ok(self, rbm, B) -> bool:
        W = rbm.W.data
        return (W.is_cuda and 1 <= B <= 64 and W.shape[1] % 4 == 0 and W.stride(0) % 4 == 0 and W.data_ptr() % 16 == 0
                and not self._groups(rbm))

    def cd_factors(self, rbm, data, cd_k, rng, data_binary=None) -> torch.Tensor:
        """The CD pass of this rank's rows; returns the factor block (a uint8 VIEW of the workspace: consume it --
        e.g. all-gather it -- before the next engine call on this RBM shape)."""
        d = self._desc(rbm, False)
        x = _f32c(data, "data")
        B, dev = x.size(0), x.device
        o = self._opts(rbm, 0.0, 0.0, cd_k)
        o.data_binary = self._hint(data, data_binary)
        sched = R.sched_cd(d.V, d.H, self._groups(rbm), cd_k)
        r, keep = self._rng(rng, sched, B, dev)
        ws = self._workspace(dev, d.V, d.H, B)
        N.check(self._lib.imdbn_rbm_cd_factors(C.byref(d), _ptr(x), x.stride(0), B, C.byref(o), C.byref(r), _ptr(ws), ws.numel(),
                                                self._stream(dev)), "imdbn_rbm_cd_factors")
        self._done(rng, r, sched)
        off, nb = self.factor_block(d.V, d.H, B)
        return ws[off:off + nb]

    def cd_factors_wire(self, rbm, data, cd_k, rng, binary: bool, next_data=None, data_binary=None) -> torch.Tensor:
        """The CD pass of this rank's rows straight into the wire form of its factor block (imdbn_rbm_cd_factors_wire =
        cd_factors + pack_factors in one call), with the next-batch hint of ``cd_step``.  Returns a reusable buffer."""
        d = self._desc(rbm, True)       # (imdbn_rbm_prefetch_ok wants the full descriptor)
        x = _f32c(data, "data")
        B, dev = x.size(0), x.device
        o = self._opts(rbm, 0.0, 0.0, cd_k)
        o.data_binary = self._hint(data, data_binary)
        sched = R.sched_cd(d.V, d.H, self._groups(rbm), cd_k)
        r, keep = self._rng(rng, sched, B, dev)
        ws = self._workspace(dev, d.V, d.H, B)
        key, nxt = self._prefetch_opts(o, d, x, next_data)
        out = self._wire_buffer("wire1", rbm, B, 1, binary)[0]
        N.check(self._lib.imdbn_rbm_cd_factors_wire(C.byref(d), _ptr(x), x.stride(0), B, C.byref(o), C.byref(r), int(bool(binary)), _ptr(out),
                                                     _ptr(ws), ws.numel(), self._stream(dev)), "imdbn_rbm_cd_factors_wire")
        self._done(rng, r, sched)
        if nxt is not None:
            self._pf[key] = (self._ident(nxt), int(o.next_slot), nxt)
        return out

    def apply_wire(self, rbm, wires: torch.Tensor, rows_per_rank, global_B, binary: bool, lr, mom):
        """The update from the gathered wire blocks (imdbn_rbm_apply_wire = unpack_factors(planes_only) + apply_factors_wire)."""
        d = self._desc(rbm, True)
        dev = wires.device
        world = int(wires.size(0))
        assert wires.dtype == torch.uint8 and wires.dim() == 2 and wires.is_contiguous()
        planes = self.gather_buffer(rbm, rows_per_rank, world)
        o = self._opts(rbm, lr, mom, 1, sparsity=getattr(rbm, "sparsity", False))
        loss = torch.empty(1, device=dev)
        N.check(self._lib.imdbn_rbm_apply_wire(C.byref(d), _ptr(wires), int(wires.stride(0)), world, int(rows_per_rank), int(global_B),
                                                int(bool(binary)), _ptr(planes), int(planes.stride(0)), C.byref(o), _ptr(loss),
                                                self._stream(dev)), "imdbn_rbm_apply_wire")
        return loss.reshape(())

    def gather_buffer(self, rbm, B, world) -> torch.Tensor:
        """Reusable [world, block bytes] uint8 buffer for the all-gather of the factor blocks."""
        W = rbm.W.data
        _, nb = self.factor_block(W.shape[0], W.shape[1], B)
        key = ("gather", W.device, W.shape[0], W.shape[1], B, world)
        buf = self._ws.get(key)
        if buf is None:
            buf = self._ws[key] = torch.empty(world, nb, dtype=torch.uint8, device=W.device)
        return buf

    # wire form of the factor block (include/imdbn_engine.h): bit-packed visible planes
    def compact_bytes(self, V, H, B, binary: bool) -> int:
        nb = C.c_size_t(0)
        N.check(self._lib.imdbn_factor_compact_bytes(int(V), int(H), int(B), int(bool(binary)), C.byref(nb)), "imdbn_factor_compact_bytes")
        return int(nb.value)

    def _wire_buffer(self, tag, rbm, B, world, binary) -> torch.Tensor:
        W = rbm.W.data
        key = (tag, W.device, W.shape[0], W.shape[1], B, world, bool(binary))
        buf = self._ws.get(key)
        if buf is None:
            # zeros: the block trailer's `bad` mark must not match a pack epoch by accident (include/imdbn_engine.h)
            buf = self._ws[key] = torch.zeros(world, self.compact_bytes(W.shape[0], W.shape[1], B, binary), dtype=torch.uint8, device=W.device)
        return buf

    def pack_factors(self, rbm, block: torch.Tensor, B, binary: bool) -> torch.Tensor:
        """This rank's factor block -> its compact wire form (a reusable buffer)."""
        W = rbm.W.data
        out = self._wire_buffer("wire1", rbm, B, 1, binary)[0]
        N.check(self._lib.imdbn_rbm_pack_factors(int(W.shape[0]), int(W.shape[1]), int(B), int(bool(binary)), _ptr(block), _ptr(out),
                                                  self._stream(W.device)), "imdbn_rbm_pack_factors")
        return out

    def compact_gather_buffer(self, rbm, B, world, binary: bool) -> torch.Tensor:
        return self._wire_buffer("wireN", rbm, B, world, binary)

    def unpack_factors(self, rbm, compact: torch.Tensor, B, binary: bool, planes_only: bool = False) -> torch.Tensor:
        """[world, compact bytes] gathered wire blocks -> [world, block bytes] full blocks for apply_factors
        (``planes_only``: just the visible planes, for apply_factors_wire)."""
        W = rbm.W.data
        world = int(compact.size(0))
        full = self.gather_buffer(rbm, B, world)
        assert compact.dtype == torch.uint8 and compact.dim() == 2 and compact.is_contiguous()
        N.check(self._lib.imdbn_rbm_unpack_factors(int(W.shape[0]), int(W.shape[1]), int(B), int(bool(binary)), _ptr(compact),
                                                    int(compact.stride(0)), world, _ptr(full), int(full.stride(0)),
                                                    int(bool(planes_only)), self._stream(W.device)), "imdbn_rbm_unpack_factors")
        return full

    def apply_factors_wire(self, rbm, wires: torch.Tensor, planes: torch.Tensor, rows_per_rank, global_B, lr, mom):
        """apply_factors with the blocks' head read from the gathered wire blocks and the visible planes from `planes`."""
        d = self._desc(rbm, True)
        dev = wires.device
        o = self._opts(rbm, lr, mom, 1, sparsity=getattr(rbm, "sparsity", False))
        loss = torch.empty(1, device=dev)
        N.check(self._lib.imdbn_rbm_apply_factors_wire(C.byref(d), _ptr(wires), int(wires.stride(0)), _ptr(planes), int(planes.stride(0)),
                                                        int(wires.size(0)), int(rows_per_rank), int(global_B), C.byref(o), _ptr(loss),
                                                        self._stream(dev)), "imdbn_rbm_apply_factors_wire")
        return loss.reshape(())

    def apply_factors(self, rbm, gathered, rows_per_rank, global_B, lr, mom):
        d = self._desc(rbm, True)
        dev = gathered.device
        o = self._opts(rbm, lr, mom, 1, sparsity=getattr(rbm, "sparsity", False))
        loss = torch.empty(1, device=dev)
        assert gathered.dtype == torch.uint8 and gathered.dim() == 2 and gathered.is_contiguous()
        N.check(self._lib.imdbn_rbm_apply_factors(C.byref(d), _ptr(gathered), int(gathered.size(0)), int(gathered.stride(0)),
                                                   int(rows_per_rank), int(global_B), C.byref(o), _ptr(loss), self._stream(dev)),
                "imdbn_rbm_apply_factors")
        return loss.reshape(())

    def apply_delta(self, rbm, packed, global_B, lr, mom, sparsity: Optional[bool] = None):
        d = self._desc(rbm, True)
        dev = packed.device
        o = self._opts(rbm, lr, mom, 1, sparsity=getattr(rbm, "sparsity", False) if sparsity is None else sparsity)
        loss = torch.empty(1, device=dev)
        N.check(self._lib.imdbn_rbm_apply_delta(C.byref(d), _ptr(packed), int(global_B), C.byref(o), _ptr(loss),
                                                 self._stream(dev)), "imdbn_rbm_apply_delta")
        return loss.reshape(())

    # ---- chains -------------------------------------------------------------------------------
    def chain(self, rbm, v_known, mask, steps: List[dict], rng, init_uniform=True, mu=None):
        d = self._desc(rbm, False)
        vk, km = _f32c(v_known, "v_known"), _f32c(mask, "mask")
        if vk.stride(0) != km.stride(0):
            vk, km = vk.contiguous(), km.contiguous()
        B, dev = vk.size(0), vk.device
        out = torch.empty(B, d.V, device=dev)
        sched = R.sched_chain(d.V, d.H, self._groups(rbm), steps, init_uniform)
        r, keep = self._rng(rng, sched, B, dev)
        mu_t = _f32c(mu, "mu") if mu is not None else None
        ws = self._workspace(dev, d.V, d.H, B)
        N.check(self._lib.imdbn_rbm_chain(C.byref(d), _ptr(vk), _ptr(km), vk.stride(0), B, int(bool(init_uniform)), len(steps),
                                           self._steps(steps), _ptr(mu_t), mu_t.stride(0) if mu_t is not None else 0,
                                           mu_t.size(1) if mu_t is not None else 0, C.byref(r), _ptr(out), out.stride(0),
                                           _ptr(ws), ws.numel(), self._stream(dev)), "imdbn_rbm_chain")
        self._done(rng, r, sched)
        return out

    def chain_pair(self, rbm, a: dict, b: dict, rng):
        """Two independent chains of `rbm` on batches of the same size as one engine call (imdbn_rbm_chain_pair); each of `a`, `b` =
        dict(v_known, mask, steps, init_uniform=True, mu=None).  Same draws, same results as ``chain(a)`` then ``chain(b)``."""
        d = self._desc(rbm, False)
        keep, specs, outs, sched = [], [], [], []
        B = dev = None
        for ch in (a, b):
            vk, km = _f32c(ch["v_known"], "v_known"), _f32c(ch["mask"], "mask")
            if vk.stride(0) != km.stride(0):
                vk, km = vk.contiguous(), km.contiguous()
            if B is None:
                B, dev = vk.size(0), vk.device
            elif vk.size(0) != B:
                raise N.EngineError("chain_pair: the two chains need the same batch size")
            steps, init_uniform = ch["steps"], bool(ch.get("init_uniform", True))
            out = torch.empty(B, d.V, device=dev)
            mu = ch.get("mu")
            mu_t = _f32c(mu, "mu") if mu is not None else None
            arr = self._steps(steps)
            sp = N.ChainSpec()
            sp.v_known, sp.mask, sp.ldk = vk.data_ptr(), km.data_ptr(), vk.stride(0)
            sp.init_uniform, sp.n_steps, sp.steps = int(init_uniform), len(steps), arr
            sp.mu, sp.ldmu, sp.Dz = (mu_t.data_ptr() if mu_t is not None else 0), (mu_t.stride(0) if mu_t is not None else 0), (mu_t.size(1) if mu_t is not None else 0)
            sp.out_v, sp.ldo = out.data_ptr(), out.stride(0)
            specs.append(sp); outs.append(out); keep.append((vk, km, mu_t, arr))
            sched += R.sched_chain(d.V, d.H, self._groups(rbm), steps, init_uniform)
        r, keep_r = self._rng(rng, sched, B, dev)
        ws = self._workspace(dev, d.V, d.H, B)
        N.check(self._lib.imdbn_rbm_chain_pair(C.byref(d), B, C.byref(specs[0]), C.byref(specs[1]), C.byref(r), _ptr(ws), ws.numel(),
                                                self._stream(dev)), "imdbn_rbm_chain_pair")
        self._done(rng, r, sched)
        return outs[0], outs[1]

    def clamped_step(self, rbm, v_known, mask, init_steps: List[dict], mu, lr, mom, cd_k, sample_h, sample_v, reclamp, rng):
        d = self._desc(rbm, True)
        vk, km = _f32c(v_known, "v_known"), _f32c(mask, "mask")
        if vk.stride(0) != km.stride(0):
            vk, km = vk.contiguous(), km.contiguous()
        B, dev = vk.size(0), vk.device
        o = self._opts(rbm, lr, mom, cd_k, sparsity=False, sample_h=sample_h, sample_v=sample_v, reclamp=reclamp)
        sched = R.sched_clamped(d.V, d.H, self._groups(rbm), init_steps, cd_k, sample_h, sample_v)
        r, keep = self._rng(rng, sched, B, dev)
        mu_t = _f32c(mu, "mu") if mu is not None else None
        loss = torch.empty(1, device=dev)
        ws = self._workspace(dev, d.V, d.H, B)
        N.check(self._lib.imdbn_rbm_clamped_step(C.byref(d), _ptr(vk), _ptr(km), vk.stride(0), B, len(init_steps),
                                                  self._steps(init_steps), _ptr(mu_t),
                                                  mu_t.stride(0) if mu_t is not None else 0,
                                                  mu_t.size(1) if mu_t is not None else 0, C.byref(o), C.byref(r),
                                                  _ptr(loss), _ptr(ws), ws.numel(), self._stream(dev)),
                "imdbn_rbm_clamped_step")
        self._done(rng, r, sched)
        return loss.reshape(())

    def clamped_stats(self, rbm, v_known, mask, init_steps: List[dict], mu, cd_k, sample_h, sample_v, reclamp, rng,
                      out: Optional[torch.Tensor] = None):
        """Data-parallel half of clamped_step: the shard's packed statistics (apply with apply_delta(sparsity=False))."""
        d = self._desc(rbm, False)
        vk, km = _f32c(v_known, "v_known"), _f32c(mask, "mask")
        if vk.stride(0) != km.stride(0):
            vk, km = vk.contiguous(), km.contiguous()
        B, dev = vk.size(0), vk.device
        o = self._opts(rbm, 0.0, 0.0, cd_k, sparsity=False, sample_h=sample_h, sample_v=sample_v, reclamp=reclamp)
        sched = R.sched_clamped(d.V, d.H, self._groups(rbm), init_steps, cd_k, sample_h, sample_v)
        r, keep = self._rng(rng, sched, B, dev)
        mu_t = _f32c(mu, "mu") if mu is not None else None
        packed = out if out is not None else torch.zeros(self.packed_floats(d.V, d.H), device=dev)
        ws = self._workspace(dev, d.V, d.H, B)
        N.check(self._lib.imdbn_rbm_clamped_stats(C.byref(d), _ptr(vk), _ptr(km), vk.stride(0), B, len(init_steps),
                                                   self._steps(init_steps), _ptr(mu_t),
                                                   mu_t.stride(0) if mu_t is not None else 0,
                                                   mu_t.size(1) if mu_t is not None else 0, C.byref(o), C.byref(r),
                                                   _ptr(packed), _ptr(ws), ws.numel(), self._stream(dev)),
                "imdbn_rbm_clamped_stats")
        self._done(rng, r, sched)
        return packed
