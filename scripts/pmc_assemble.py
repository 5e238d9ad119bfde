"""Launch the library's assembly kernel a few times (1024 x 10 and 65536 x 10) for a rocprofv3 counter pass:

    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmcw -- python3 scripts/pmc_assemble.py
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmcf -- python3 scripts/pmc_assemble.py

(separate passes: the two counters do not fit one TCC pass, MI355X_MICROARCH.md "rocprofv3 PMC slots")."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mrs_uav_trajectory_generation_amd import api

ctx = api.Context(0)
ctx.use_torch_stream()
for P in (1024, 65536):
    S = 10
    so = (np.arange(P + 1) * S).astype(np.int32)
    plan = api.Plan(ctx, so)
    t = torch.rand(P * S, dtype=torch.float64, device="cuda") * 5 + 0.5
    H = torch.empty(plan.block_doubles, dtype=torch.float64, device="cuda")
    A = torch.empty_like(H)
    for _ in range(5):
        plan.assemble(4, t, H, A)
    torch.cuda.synchronize()
    plan.close()
