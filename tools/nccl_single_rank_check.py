"""World size 1 over RCCL on one GPU: the async gather of gather_bands_begin / gather_bands_end against the plain batch call
(what the collective's stream semantics can be checked with on a one-GPU box)."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, parallel
from raytracer_amd.scene import FlatScene

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
ctx = capi.Context(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = ctx.upload(FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(640, 360))
p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=3, sub_sample=1)
eng = parallel.DeviceEngine(sc, p, device="cuda", with_lists=False)
ref = parallel.render_sharded_batch(eng, 640, 360, 3).clone()
# force the collective path although there is one rank: gather to self
parts = parallel.trace_bands_batch(eng, 640, 360, 3)
got = [torch.empty_like(parts)]
w = dist.gather(parts.contiguous(), got, dst=0, async_op=True)
parts2 = parallel.trace_bands_batch(eng, 640, 360, 3)          # traced while the gather runs
w.wait(); torch.cuda.current_stream().synchronize()
assert torch.equal(got[0], ref) and torch.equal(parts2, ref), "async gather differs"
print("nccl single-rank async gather: OK")
dist.destroy_process_group()
