import csv,glob,numpy as np,sys
f=sorted(glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True))[-1]
rows=[r for r in csv.DictReader(open(f)) if 'imdbn' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
seq=[('K1' if 'k1_stream' in r['Kernel_Name'] else 'K2' if 'k2_stream' in r['Kernel_Name'] else 'K3' if 'assoc' in r['Kernel_Name'] else 'o',(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Kernel_Name'][13:33]) for r in rows]
pos=[d for i,(k,d,n) in enumerate(seq) if k=='K1' and i>0 and seq[i-1][0]=='K3']
neg=[d for i,(k,d,n) in enumerate(seq) if k=='K1' and i>0 and seq[i-1][0]=='K2']
k2=[d for k,d,n in seq if k=='K2']; k3=[d for k,d,n in seq if k=='K3']
print('   K1pos median %.2f  K1neg median %.2f  K2 %.2f  K3 %.2f  sum %.2f'%(np.median(pos),np.median(neg),np.median(k2),np.median(k3),np.median(pos)+np.median(neg)+np.median(k2)+np.median(k3)))
