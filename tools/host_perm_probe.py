import sys, time, numpy as np, ctypes as C
sys.path.insert(0,'/root/repo')
import __graft_entry__ as ge
pkg=ge.load_package(); lib=pkg.lib(); N=pkg._native
n=20000
x=np.arange(12*n,dtype=np.uint64).reshape(n,12); out=np.zeros_like(x)
lib.p2mt_host_poseidon_permute(N.ptr(x),N.ptr(out),n)
t=time.perf_counter(); lib.p2mt_host_poseidon_permute(N.ptr(x),N.ptr(out),n); dt=time.perf_counter()-t
print(sys.argv[1] if len(sys.argv)>1 else "auto", "us/perm (independent inputs)", dt/n*1e6, int(out.sum()%1000003))
