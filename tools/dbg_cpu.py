import os, sys, time, torch
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), "torch threads", torch.get_num_threads(), flush=True)
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
except Exception as e:
    print("no cgroup v2 cpu.max", e, flush=True)
n = int(sys.argv[1])
torch.set_num_threads(n)
a = torch.randn(4112, 512); w = torch.randn(2048, 512)
t0 = time.time()
for _ in range(20): torch.nn.functional.linear(a, w)
print("threads", n, "20 linears", time.time() - t0, flush=True)
