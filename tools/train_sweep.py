"""dev: cfg-2 step time (graph replays) under the environment it is started with."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tools.quick_bench import timeit
dev = torch.device("cuda", 0)
st, _ = bench.build_workload(dev, 0)
for _ in range(5):
    st.run()
print(f"{timeit(st.run, iters=300, warm=30):.1f}")
