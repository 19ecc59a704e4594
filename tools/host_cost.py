import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from rebvio_amd import synth, backend as B
W,H=640,480
frames, cam = synth.render_stream(W,H,8)
ctx = B.Context(B.default_params(H,W,fm=cam.fm,cx=cam.cx,cy=cam.cy,keylines_ref=15000,keylines_max=16000, map_pool=8))
dev = ctx.upload_frames(frames); npx=W*H
# warm
for i in range(30):
    m = ctx.detect_u8_device(dev + (i%8)*npx, i*50000); m.release()
torch.cuda.synchronize()
N=200
t0=time.perf_counter()
for i in range(N):
    m = ctx.detect_u8_device(dev + (i%8)*npx, i*50000); m.release()
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print("detect-only: host enqueue %.1f us/frame, total incl GPU drain %.1f us/frame" % ((t1-t0)/N*1e6, (t2-t0)/N*1e6))
# full pipeline host time
for i in range(60): ctx.push_frame_u8_device(dev + (i%8)*npx, i*50000)
torch.cuda.synchronize()
t0=time.perf_counter()
for i in range(N): ctx.push_frame_u8_device(dev + (i%8)*npx, i*50000)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("push_frame: %.1f us/frame (host loop), %.1f incl drain" % ((t1-t0)/N*1e6,(t2-t0)/N*1e6))
