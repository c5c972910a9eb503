import os, sys
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.training import CLVOTrainer
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
tr = CLVOTrainer(syn.to_torch(syn.make_clvo_state(seed=1)), 2, 2, device=dev)
fl = torch.from_numpy(syn.make_flow(4, 376, 1232, seed=5)).view(2, 2, 2, 376, 1232).to(dev)
r = np.random.RandomState(0)
loss, _, _ = tr.forward_backward(fl, torch.from_numpy(r.uniform(-.02, .02, (2, 2, 3)).astype(np.float32)), torch.from_numpy(r.uniform(-.5, 1.5, (2, 2, 3)).astype(np.float32)))
before = tr.grads.clone()
dist.all_reduce(tr.grads)      # RCCL collective straight on the library-owned gradient buffer (world size 1: identity)
torch.cuda.synchronize()
print("loss", loss, "grad norm", float(before.norm()), "view intact after all_reduce:", bool(torch.equal(before, tr.grads)),
      "device ptr matches:", tr.grads.data_ptr() != 0, "numel", tr.grads.numel())
g = tr.gradient("lstm1.weight_hh")
print("named read agrees with the view:", float(g.abs().sum()) > 0)
dist.destroy_process_group()
