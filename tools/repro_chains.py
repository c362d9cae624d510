"""Dev tool: repeat the half-batch-chains comparison on the ImageNet-64-width model and report which runs differ (run-to-run determinism of each mode too)."""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_parity import _uvit, load_config, REPO
from duodiff_amd import _lib as L
from duodiff_amd.engine import sample_loop
B, S, C_, steps, tsw = 256, 64, 3, 4, 2
cfg = load_config(REPO / "configs" / "uvit_imagenet64_3.yaml")
m_s, _ = _uvit(cfg, 31, "bf16", max_batch=B)
m_f, mp_f = _uvit(cfg, 32, "bf16", max_batch=B)
es, ef = m_s.engine_model(B), m_f.engine_model(B)
ctx = es.ctx
x0 = torch.randn(B, C_, S, S, generator=torch.Generator().manual_seed(4)).cuda()
y = torch.randint(0, int(mp_f.num_classes), (B,), generator=torch.Generator().manual_seed(5)).cuda()
stream = torch.cuda.Stream(); stream.wait_stream(torch.cuda.current_stream())
ref = {}
with torch.cuda.stream(stream):
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        for name, flags in (("chained", 0), ("single", L.DD_DEV_NO_CHAINS)):
            ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, flags))
            x = x0.clone()
            sample_loop(ctx, es, ef, x, t_switch=tsw, t_start=999, t_end=1000 - steps, y=y, seed=9, noise="philox", use_graph=True, stream=stream)
            stream.synchronize()
            if name not in ref: ref[name] = x.clone()
            d = (x != ref[name]).flatten(1).any(1).nonzero().flatten().tolist()
            dc = (x != ref["chained"]).flatten(1).any(1).nonzero().flatten().tolist()
            print(f"rep {rep} {name}: images differing from first {name} run: {d[:8]} ({len(d)}); from first chained run: {dc[:8]} ({len(dc)})", flush=True)
ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, 0))
