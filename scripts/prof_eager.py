import cProfile, pstats, sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
sys.argv = [sys.argv[0]]
a = bench.parse(); a.mode = "eager"; a.layers = 32
dev = torch.device("cuda:0")
w = bench.Workload(a, dev)
for _ in range(5): w.step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): w.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
