import sys, time, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo')
import torch
from rebvio_amd import synth, backend as B
W,H=640,480
NB=24
frames, cam = synth.render_stream(W,H,NB)
ctx = B.Context(B.default_params(H,W,fm=cam.fm,cx=cam.cx,cy=cam.cy,keylines_ref=15000,keylines_max=16000))
dev = ctx.upload_frames(frames); npx=W*H
order = synth.pingpong_indices(NB, 5000)
L = B.lib(); out = B.PairOut(); n = C.c_int()
def push(i):
    return L.rebvio_hip_push_frame_u8_device(ctx.h, C.c_void_p(dev + int(order[i])*npx), i*50000, C.byref(out), C.byref(n))
k=0
for _ in range(80): push(k); k+=1
torch.cuda.synchronize()
N=1500
t0=time.perf_counter()
for _ in range(N): push(k); k+=1
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("push_frame (ping-pong, tracking valid, klm=%d): %.1f us/frame host loop, %.1f incl drain -> %.0f fps" % (out.klm_num, (t1-t0)/N*1e6,(t2-t0)/N*1e6, N/(t2-t0)))
# detect only
ctx.flush()
ms=[]
t0=time.perf_counter()
for i in range(300):
    m = ctx.detect_u8_device(dev + int(order[i])*npx, i*50000); m.release()
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("detect-only: host %.1f us/frame, incl drain %.1f" % ((t1-t0)/300*1e6,(t2-t0)/300*1e6))
