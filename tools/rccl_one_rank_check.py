import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/tests/golden"]
import numpy as np, torch, torch.distributed as dist
from synth_weights import make_state_dict, synth_pcm_i16
from uit_mobile_amd import models
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29777")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sd = make_state_dict("uit_xxxs", 1)
m = models.uit_xxxs(outputdim=537, target_length=102); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.cuda().eval()
x = torch.from_numpy(synth_pcm_i16(8, 16000, seed=1)).cuda()
want = m(x).clone()
st = m.frontend(x)
slots = m.amax_tensor(x.device)
before = slots.clone()
w = dist.all_reduce(slots, op=dist.ReduceOp.MAX, async_op=True)    # RCCL on the library-owned device slots (foreign pointer)
w.wait()
p = m.encode(st)
out = torch.empty((8, 537), device="cuda")
dist.all_gather_into_tensor(out, p)
torch.cuda.synchronize()
print("slots unchanged:", bool(torch.equal(before, slots)), " gathered == plain forward:", bool(torch.equal(out, want)))
dist.destroy_process_group()
