import sys, time, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import torch, numpy as np
from skrample_amd import _hip as H
lib = H.load(); print(lib.skr_build_info().decode())
dev = torch.device('cuda:0')
print(torch.cuda.get_device_name(0))
def mkplan(coefs, dt=H.BF16, out=H.BF16, zeta=0.0, stream=0, sample_numel=0, noise=0):
    p = H.StepPlanC(); p.n_terms=len(coefs); p.n_group_a=len(coefs); p.dtype_a=dt; p.dtype_b=dt; p.out0_dtype=out; p.out1_dtype=H.NONE
    p.acc_f64=0; p.noise_mode=noise; p.zeta0=zeta; p.stream0=stream; p.sample_numel=sample_numel
    for i,c in enumerate(coefs): p.coef0[i]=c
    return p
# correctness: DPM-2 like 4-term lincomb bf16
B,C,Hh,W = 64,4,128,128
g = torch.Generator(device=dev).manual_seed(0)
ins = [torch.randn(B,C,Hh,W, device=dev, generator=g).bfloat16() for _ in range(4)]
coefs = [1.0123, -0.5321, 0.1234, 0.4321]
out = torch.empty_like(ins[0])
H.launch_step(mkplan(coefs), ins, out, None, None, out.numel(), dev)
ref = sum(c*t.float() for c,t in zip(coefs, ins))
torch.cuda.synchronize()
err = (out.float()-ref).abs().max().item(); print("max abs err vs fp32 ref (bf16 out):", err, "ulp-ish", (out.float()-ref.bfloat16().float()).abs().max().item())
# fp32 out
out32 = torch.empty(B,C,Hh,W, device=dev)
H.launch_step(mkplan(coefs, out=H.F32), ins, out32, None, None, out.numel(), dev); torch.cuda.synchronize()
print("fp32-out rel err:", ((out32-ref).abs().max()/ref.abs().max()).item())
# philox parity
from skr_oracle import noise as ON
u = torch.empty(4*1000, dtype=torch.int32, device=dev)
H.check(lib.skr_philox_u32(u.data_ptr(), 0x123456789abcdef, 7, 5, 1000, H.current_stream_ptr(dev)), "philox"); torch.cuda.synchronize()
blocks = np.arange(5, 1005, dtype=np.uint64)
ctr = np.stack([(blocks & 0xffffffff).astype(np.uint32), (blocks>>np.uint64(32)).astype(np.uint32), np.full(1000,7,np.uint32), np.zeros(1000,np.uint32)], -1)
key = np.array([0x123456789abcdef & 0xffffffff, 0x123456789abcdef>>32], dtype=np.uint32)
exp = ON.philox4x32(ctr, key).reshape(-1)
print("philox u32 bit-exact:", np.array_equal(u.cpu().numpy().view(np.uint32), exp))
# normals
seeds = torch.tensor([42+i for i in range(4)], dtype=torch.int64, device=dev)
z = torch.empty(4, 4*64*64, device=dev)
H.check(lib.skr_noise_random(z.data_ptr(), H.F32, seeds.data_ptr(), 3, 4, 4*64*64, H.current_stream_ptr(dev)), "rand"); torch.cuda.synchronize()
zo = np.stack([ON.philox_normal(42+i, 3, 4*64*64) for i in range(4)])
d = np.abs(z.cpu().numpy()-zo); print("normal max abs diff vs oracle:", d.max(), "mean", z.mean().item(), "std", z.std().item())
# timing: DPM-2 SDE at B=256: 4 inputs bf16 + philox noise -> bf16
def bench(B, noise, nsets=4, iters=200, K=4):
    n = B*4*128*128
    sets = [([torch.randn(n, device=dev).bfloat16() for _ in range(K)], torch.empty(n, device=dev, dtype=torch.bfloat16)) for _ in range(nsets)]
    seeds = torch.arange(B, dtype=torch.int64, device=dev)+42
    p = mkplan([1.01,-0.53,0.12,0.43][:K], zeta=0.3 if noise else 0.0, stream=1, sample_numel=4*128*128, noise=1 if noise else 0)
    for i in range(20): H.launch_step(p, sets[i%nsets][0], sets[i%nsets][1], None, seeds, n, dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0=time.perf_counter(); e0.record()
    for i in range(iters): H.launch_step(p, sets[i%nsets][0], sets[i%nsets][1], None, seeds, n, dev)
    e1.record(); torch.cuda.synchronize(); t1=time.perf_counter()
    ms = e0.elapsed_time(e1)/iters
    gb = n*(2*K+2)/1e9
    print(f"B={B} K={K} noise={noise}: {ms*1e3:.1f} us/step (event), wall {(t1-t0)/iters*1e6:.1f} us, {gb/ms*1e3/1e3:.2f} TB/s algorithmic, frac of 8TB/s {gb/ms/8:.3f}")
for B in (64, 256):
    for noise in (False, True):
        bench(B, noise)
bench(256, False, K=2); bench(256, False, K=8)
# copy ceiling
a = torch.randn(256*4*128*128*2, device=dev).bfloat16(); b = torch.empty_like(a)
for _ in range(5): b.copy_(a)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(50): b.copy_(a)
e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/50; print("torch copy_:", a.numel()*4/ms/1e9, "TB/s")
