import sys
sys.path.insert(0, '/root/repo')
from gpy_amd import _lib as L
ks=[512]
out=[]
for nt in range(24,129,8):
    ms=L.dbg_update_nt(nt,ks,8)
    T=nt*(nt+1)//2
    fl=T*128*128*2*512
    out.append("nt=%d T=%d rem=%d: %.3f ms %.1f TF"%(nt,T,T%512,ms[0],fl/ms[0]/1e9))
print("\n".join(out))
