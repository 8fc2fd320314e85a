"""tools/cpu_forward_scaling.py THREADS [mallopt]: rows/s of the matrix-product fp32 forward (bench.py's cpu_baseline, MMForward) with THREADS row-parallel workers of one MKL
thread each, and on one thread; prints the host's CPU budget first (cgroup quota, affinity).  Round 6: why 64 workers on the GPU box's host give 9x one thread, not 64x."""
import sys, time, ctypes, numpy as np, torch, os
import torch.nn.functional as F
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0,'/root/repo')
import omok_ai_amd as oa
if len(sys.argv) > 2 and sys.argv[2] == 'mallopt':
    libc = ctypes.CDLL("libc.so.6")
    print('mallopt', libc.mallopt(-3, 1 << 30), libc.mallopt(-1, 1 << 30))  # M_MMAP_THRESHOLD, M_TRIM_THRESHOLD
n=15; hw=225
tensors = oa.weights.init_random(n, seed=0)
sh = oa.weights.tensor_shapes(n)
t = [torch.as_tensor(np.asarray(x, dtype=np.float32).reshape(s_)) for x, s_ in zip(tensors, sh)]
w_in,b_in=t[0].reshape(3,128).contiguous(),t[1]
blocks=[]
for i in range(3):
    w0,b0,dw,pw,b1,w2,b2=t[2+7*i:9+7*i]
    blocks.append((w0.reshape(128,32).contiguous(),b0,dw.reshape(9,32).contiguous(),pw.reshape(32,32).contiguous(),b1,w2.reshape(32,128).contiguous(),b2))
fc=t[23:31]
def chunk(x):
    b=x.shape[0]
    a=F.leaky_relu_(torch.addmm(b_in,x.reshape(b*hw,3),w_in),0.2)
    for w0,b0,dw,pw,b1,w2,b2 in blocks:
        h=F.leaky_relu_(torch.addmm(b0,a,w0),0.2)
        hp=F.pad(h.view(b,n,n,32),(0,0,1,1,1,1))
        d=hp[:,0:n,0:n,:]*dw[0]
        for tap in range(1,9): d.addcmul_(hp[:,tap//3:tap//3+n,tap%3:tap%3+n,:],dw[tap])
        g=F.leaky_relu_(torch.addmm(b1,d.reshape(b*hw,32),pw),0.2)
        a=F.leaky_relu_(torch.addmm(b2,g,w2).add_(a),0.2)
    h0=F.leaky_relu_(torch.addmm(fc[1],a.view(b,hw*128),fc[0]),0.2)
    h1=F.leaky_relu_(torch.addmm(fc[3],h0,fc[2]),0.2)
    return torch.softmax(torch.addmm(fc[7],h1,fc[6]),dim=1),torch.tanh(torch.addmm(fc[5],h1,fc[4]))
T=int(sys.argv[1])
try: print('cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as ex: print('cpu.max', ex)
print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count(), 'loadavg', os.getloadavg())
x=torch.from_numpy((np.random.rand(4096,3*hw)<0.2).astype(np.float32))
torch.set_num_threads(1)
pool=ThreadPoolExecutor(T, initializer=lambda: torch.set_num_threads(1))
for cs in (64,128):
    starts=list(range(0,4096,cs))
    with torch.no_grad():
        list(pool.map(lambda i: chunk(x[i:i+cs]), starts))
        t0=time.perf_counter(); list(pool.map(lambda i: chunk(x[i:i+cs]), starts)); dt=time.perf_counter()-t0
    print(T, cs, f"{4096/dt:.0f} rows/s")
with torch.no_grad():
    chunk(x[:64]); t0=time.perf_counter(); [chunk(x[i:i+64]) for i in range(0,512,64)]; print('1 thread', 512/(time.perf_counter()-t0))
