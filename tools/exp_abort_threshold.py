"""How many compute units may a foreign kernel hold before a rolling launch (5 entries of 32k) cannot make progress within the wait limit?
(round 5: sizing tests/test_rolling_gpu.py::test_an_abandoned_rolling_launch_...).  One child process per setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = """
import sys, time, torch
from fastkv_amd import ops
from fastkv_amd._lib import load
L = load(); dev = torch.device('cuda:0')
B, H, Hkv, S, D = 5, 32, 8, 32768, 128
g = torch.Generator(device=dev).manual_seed(99)
q = torch.randn(B, S, H, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
k = torch.randn(B, S, Hkv, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
v = torch.randn(B, S, Hkv, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
def run():
    out = ops.update_kv(q, k, v, 8, 7, 'avgpool', 2048, 2048, 'score', return_indices=True, return_scores=True)
    torch.cuda.current_stream().synchronize()
    return out
run(); run()
side = torch.cuda.Stream()
assert L.fastkv_debug_occupy(HELD, LDS * 1024, 900 * 1000, side.cuda_stream) == 0
time.sleep(0.02)
t0 = time.perf_counter()
run()
dt = (time.perf_counter() - t0) * 1e3
print('held', HELD, 'lds KiB', LDS, 'call took %.1f ms' % dt, 'status', L.fastkv_last_status(), flush=True)
torch.cuda.synchronize()
"""
for held, lds in ((200, 128), (232, 128), (240, 128), (244, 128), (248, 128), (256, 100), (256, 120)):
    env = dict(os.environ, FASTKV_SPIN_LIMIT_MS="40", FASTKV_STRICT_PLACEMENT="0")
    r = subprocess.run([sys.executable, "-c", f"HELD = {held}\nLDS = {lds}\n" + CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    print((r.stdout.strip().splitlines() or ["?"])[-1], "| rc", r.returncode, r.stderr.strip().splitlines()[-1][:200] if r.returncode else "", flush=True)
