"""GPU: the C-level gather (include/dsp_amd.h dsp_gather_*, SURVEY 8e: "RCCL over xGMI used only to gather the per-clip feature
vectors") for a host that drives the GPUs of a node from ONE process.  A one-GPU box can check what does not need a second GPU:
RCCL resolved by dlopen, ncclCommInitAll, the grouped all-gather on the caller's stream behind the MFCC kernel of its batch, slot reuse
with the next batch computed while the gather of the previous one is in flight, the results.  The xGMI transfer itself needs the
driver's 8-GPU node.  Runs in a child process and keeps the two-phase rule of the torch.distributed test: a box whose RCCL cannot
bring up a one-rank communicator (before any code of this library runs a collective) skips; after "RCCL-UP" every failure fails."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_c_gather_on_one_rank():
    code = r"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import dsp_amd
from dsp_amd import lib as L
lib = L.load()
devs = (C.c_int * 1)(0)
g = C.c_void_p()
rc = lib.dsp_gather_create(devs, 1, C.byref(g))
if rc < 0:
    print("RCCL-DOWN", L.last_error(), flush=True); sys.exit(0)
probe = torch.arange(64, dtype=torch.float32, device="cuda"); out = torch.zeros_like(probe)
send = (C.c_void_p * 1)(probe.data_ptr()); recv = (C.c_void_p * 1)(out.data_ptr())
rc = lib.dsp_gather_all(g, send, recv, probe.numel() * 4, None)
torch.cuda.synchronize()
if rc < 0 or not torch.equal(out, probe):
    print("RCCL-DOWN", L.last_error(), flush=True); sys.exit(0)
print("RCCL-UP", flush=True)
# phase 2: batches of per-clip MFCC matrices, the gather of batch k on its own stream while batch k + 1 is computed
assert lib.dsp_gather_n_devices(g) == 1
plan = dsp_amd.MfccPlan(dsp_amd.default_config())
gen = torch.Generator(device="cuda").manual_seed(5)
batches = [torch.rand((64, 16000), device="cuda", generator=gen) * 2 - 1 for _ in range(5)]
want = [plan.clips(b, 500).clone() for b in batches]
comp, comm = torch.cuda.Stream(), torch.cuda.Stream()
local = [torch.empty((64, 98, 13), device="cuda") for _ in range(2)]
gathered = [torch.zeros((1, 64, 98, 13), device="cuda") for _ in range(2)]
done = [torch.cuda.Event() for _ in range(2)]
got = []
for k, b in enumerate(batches):
    s = k % 2
    with torch.cuda.stream(comp):
        comp.wait_event(done[s])                      # the gather that last read this slot
        plan.clips(b, 500, local[s])
        ready = torch.cuda.Event(); ready.record(comp)
    comm.wait_event(ready)
    send = (C.c_void_p * 1)(local[s].data_ptr()); recv = (C.c_void_p * 1)(gathered[s].data_ptr())
    streams = (C.c_void_p * 1)(comm.cuda_stream)
    L.check(lib.dsp_gather_all(g, send, recv, local[s].numel() * 4, streams), "dsp_gather_all")
    done[s].record(comm)
    if k >= 1:
        p = (k - 1) % 2
        done[p].synchronize()
        got.append(gathered[p][0].clone())
torch.cuda.synchronize()
got.append(gathered[(len(batches) - 1) % 2][0].clone())
for a, w in zip(got, want):
    assert torch.equal(a, w)
# argument checks
assert lib.dsp_gather_all(g, None, recv, 4, None) < 0
two = (C.c_int * 2)(0, 0); h = C.c_void_p()
assert lib.dsp_gather_create(two, 2, C.byref(h)) < 0          # a device listed twice
lib.dsp_gather_destroy(g)
print("C-GATHER-ONE-RANK-OK")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    if "RCCL-UP" not in r.stdout:
        pytest.skip("RCCL could not bring up a one-rank communicator on this box: " + (r.stdout.strip().splitlines() + r.stderr.strip().splitlines() or ["?"])[-1][:300])
    assert "C-GATHER-ONE-RANK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
