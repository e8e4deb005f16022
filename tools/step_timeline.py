import csv,glob,numpy as np,sys
f=sorted(glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True))[-1]
n=int(sys.argv[2]) if len(sys.argv)>2 else 20
rows=[r for r in csv.DictReader(open(f)) if 'imdbn' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
k3=[(int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in rows if 'assoc_update' in r['Kernel_Name']]
k1=[(int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in rows if 'k1_stream' in r['Kernel_Name']]
ends=np.array([e for s,e in k3])
d=np.diff(ends)/1e3
print("last %d step durations (K3 end to K3 end, us):"%n, np.round(d[-n:],1))
# first kernel of the last n steps: K1pos start of step -n
first_k1=k1[-2*n][0]
print("timed region on the GPU: first K1 start -> last K3 end: %.1f us = %.2f us/step"%((ends[-1]-first_k1)/1e3,(ends[-1]-first_k1)/1e3/n))
print("gap before the first timed K1 (from previous K3 end): %.1f us"%((first_k1-ends[-n-1])/1e3))
