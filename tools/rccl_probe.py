#!/usr/bin/env python3
"""GPU-box probe: does the RCCL ("nccl") backend run here with (a) one rank, (b) two ranks sharing device 0?
Prints one line per case.  (An 8-GPU node is the driver's to launch; this shows what a 1-GPU box can execute.)"""
import os
import subprocess
import sys

WORKER = r"""
import os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
x = torch.full((1 << 20,), rank + 1, dtype=torch.uint8, device="cuda:0")
outs = [torch.empty_like(x) for _ in range(world)] if rank == 0 else None
dist.gather(x, outs, dst=0)
torch.cuda.synchronize()
if rank == 0:
    print("GATHER_OK", [int(o[0]) for o in outs])
dist.destroy_process_group()
"""


def run(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + world), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", WORKER], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=180)[0].decode())
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append("TIMEOUT")
    ok = any("GATHER_OK" in o for o in outs)
    print(f"world={world}: {'OK' if ok else 'FAILED'}")
    if not ok:
        for o in outs:
            print("   | " + "\n   | ".join(o.strip().splitlines()[-6:]))


if __name__ == "__main__":
    run(1)
    run(2)
