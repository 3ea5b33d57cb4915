"""Probe: does RCCL accept two ranks on ONE device (it would let a 1-GPU box run the N > 1 exchange for real)?"""
import os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.full((4,), float(rank), dtype=torch.float64, device="cuda")
    out = torch.empty(8, dtype=torch.float64, device="cuda")
    dist.all_gather_into_tensor(out, x)
    torch.cuda.synchronize()
    print("rank", rank, "gathered", out.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print("rank", rank, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
    sys.exit(3)
