import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet
class O:
    model = {'out_c': 3}
torch.manual_seed(0)
m = HighResolutionNet(O()).cuda().eval()
x = torch.rand((4, 3, 512, 512), device='cuda')
with torch.no_grad():
    for _ in range(3):
        m(x)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with torch.no_grad(), profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m(x)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by='self_cuda_time_total', row_limit=12, max_name_column_width=40, max_src_column_width=110))
