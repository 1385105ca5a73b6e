"""cProfile of the HOST side of bench.py's step (default schedule), 200 steps: where do the 0.46 ms of enqueue time per step go?"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
work = bench.HotPathPrefill(dev, seed=1)
work.defer = os.environ.get("EXP_DEFER", "1") == "1"
for _ in range(5):
    work.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    work.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
