import ctypes, torch, sys
sys.path.insert(0, "/root/repo")
from aicity_action_amd import _hip
torch.zeros(1, device="cuda:0")
L = ctypes.CDLL("/root/repo/aicity_action_amd/lib/libmvit_hip.so")
print("occupancy fwd", L.mvit_internal_skip_pool_occupancy(0), "bwd", L.mvit_internal_skip_pool_occupancy(1))
p = torch.cuda.get_device_properties(0)
print(p)
