import sys, time, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.config import ModelParams, load_config
from duodiff_amd.weights import synthetic_state_dict
from duodiff_amd.uvit import UViT
from duodiff_amd.engine import sample_loop
mode = sys.argv[1]
mp = ModelParams.from_dict(load_config("/root/repo/configs/uvit_celeba_3.yaml"))
m = UViT(**mp.as_dict(), precision="bf16", max_batch=128).load_state_dict(synthetic_state_dict(mp, 1)).to("cuda")
em = m.engine_model(128)
x = torch.randn(128, 3, 64, 64).cuda()
st = torch.cuda.Stream()
print("start", mode, flush=True)
t0 = time.time()
with torch.cuda.stream(st):
    sample_loop(em.ctx, em, None, x, t_start=999, t_end=990, seed=1, noise="philox", use_graph=(mode == "graph"), stream=st)
    print("enqueued", time.time() - t0, flush=True)
    st.synchronize()
print("done", mode, time.time() - t0, float(x.abs().max()), flush=True)
