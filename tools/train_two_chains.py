"""Training step as two concurrent half-batch chains (two engines on one parameter set, two streams, gradients summed) vs the
one-engine step: does the sampler's two-chain trick carry over to training?   gpurun -- python tools/train_two_chains.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smd_amd.ncsn as N
import smd_amd.schedule as S
from smd_amd.engine import NetConfig, Engine

dev = "cuda:0"
cfg = NetConfig(architecture="TransformerDDPM", data_channels=512, seq_len=32, num_timesteps=1000)
betas = S.create_noise_schedule(1e-6, 0.01, 1000, "linear")
B = 256
g = torch.Generator().manual_seed(1234)
x0 = torch.clamp(0.25 * torch.randn(B, 32, 512, generator=g), -1, 1).to(dev)


def timeit(f, steps=30, warm=5):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


# ---- one engine, B = 256
model = N.Model(cfg, dev, seed=0)
eng = model.train_engine(ema=False)
eng.set_schedule(betas, with_sampler=False)
eng.bind(B, training=True)
step_id = [0]


def one():
    eng.loss_backward(x0, None, None, seed=step_id[0], stage=0)
    eng.optimizer_step(1e-3, 0.98, 10000, 1.0, 0.999, 1.0)
    step_id[0] += 1


t_one = sorted(timeit(one) for _ in range(3))[1]
g_one = eng.grads.clone()

# ---- two engines, B = 128 each, one parameter set
model2 = N.Model(cfg, dev, seed=0)
e1 = model2.train_engine(ema=False)
e2 = Engine(cfg, dev, share_params_with=model2.engine)
e2.enable_training(False)
for e in (e1, e2):
    e.set_option("nt256_min_tiles", 128)
    e.set_schedule(betas, with_sampler=False)
    e.bind(B // 2, training=True)
mode = os.environ.get("STREAMS", "fresh")
if mode == "fresh":
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
else:                      # chain 1 on the current stream, chain 2 on one new stream
    s1, s2 = torch.cuda.current_stream(), torch.cuda.Stream()
xa, xb = x0[:B // 2].contiguous(), x0[B // 2:].contiguous()
step2 = [0]


def two():
    cur = torch.cuda.current_stream()
    if s1 is not cur:
        s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        e1.loss_backward(xa, None, None, seed=step2[0], sample_offset=0, global_batch=B, stage=0)
    with torch.cuda.stream(s2):
        e2.loss_backward(xb, None, None, seed=step2[0], sample_offset=B // 2, global_batch=B, stage=0)
    if s1 is not cur:
        cur.wait_stream(s1)
    cur.wait_stream(s2)
    e1.grads.add_(e2.grads)
    e1.optimizer_step(1e-3, 0.98, 10000, 1.0, 0.999, 1.0)
    step2[0] += 1


t_two = sorted(timeit(two) for _ in range(3))[1]
# same first-step gradient?  (fresh models, step 0)
m3 = N.Model(cfg, dev, seed=0); ea = m3.train_engine(ema=False); ea.set_schedule(betas, with_sampler=False); ea.bind(B, training=True)
ea.loss_backward(x0, None, None, seed=0, stage=0)
m4 = N.Model(cfg, dev, seed=0); eb = m4.train_engine(ema=False); ec = Engine(cfg, dev, share_params_with=m4.engine); ec.enable_training(False)
for e, xs, off in ((eb, xa, 0), (ec, xb, B // 2)):
    e.set_option("nt256_min_tiles", 128); e.set_schedule(betas, with_sampler=False); e.bind(B // 2, training=True)
    e.loss_backward(xs, None, None, seed=0, sample_offset=off, global_batch=B, stage=0)
torch.cuda.synchronize()
gs = eb.grads + ec.grads
rel = float((gs - ea.grads).norm() / ea.grads.norm())
print(f"one engine B=256: {t_one * 1e3:.0f} us per train step ({1e3 / t_one:.1f} steps/s)")
print(f"[STREAMS={mode} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '-')}] two engines B=128 on two streams + gradient sum: {t_two * 1e3:.0f} us per train step ({1e3 / t_two:.1f} steps/s)")
print(f"first-step gradient, sum of the halves vs full batch: rel {rel:.2e}")
