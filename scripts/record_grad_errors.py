"""Records what the training kernels ACHIEVE against their two references - the reference's own autograd (tests/golden/*_grads.npz)
and the CPU oracle's autograd - per case and per parameter tensor, into tests/golden/achieved_grad_errors.json.  The parity tests of
tests/test_gpu_train.py assert <= 10 x these numbers.  GPU box:   python scripts/record_grad_errors.py [out.json]
Re-record (and commit the file) when a kernel change moves the errors on purpose; a test failing against the recorded numbers is a
regression until shown otherwise."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for q in (os.path.join(ROOT, "oracle"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, q)
import torch  # noqa: E402
import test_gpu_train as TT  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "achieved_grad_errors.json")
rec = {"reference": {}, "oracle": {}, "reference_norm": {}}
for name in TT.GRAD_CASES:
    loss, ref, err, nerr, spread = TT.reference_case_errors(name)
    # (the norm check shares the bar: keep the larger of the two per tensor)
    rec["reference"][name] = {k: max(err[k], nerr[k]) for k in err}
    rec["reference_norm"][name] = nerr
    print(name, "loss", loss, ref, "worst", max(err.values()), flush=True)
rec["config2_forward"] = {}
for name in ("full_train_grads_8192", "full_train_grads_8192_w4"):
    if os.path.exists(os.path.join(ROOT, "tests", "golden", name + ".npz")):
        loss, ref, fwd, z, err, nerr = TT.config2_errors(name)
        rec["reference"][name] = {k: max(err[k], nerr[k]) for k in err}
        rec["config2_forward"][name] = {"loss": loss, "reference_loss": ref, "max_abs": fwd}
        print(name, "loss", loss, ref, fwd, "worst", max(err.values()), flush=True)
for name, nrays, nsamp in TT.ORACLE_CASES:
    err = TT.oracle_case_errors(name, nrays, nsamp)
    rec["oracle"][TT.oracle_case_key(name, nrays, nsamp)] = err
    print(name, nrays, nsamp, "worst", max(err.values()), flush=True)
try:
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    head = "(no git on the box)"
rec["meta"] = {"device": torch.cuda.get_device_name(0), "head": head, "what": "relative L2 error per parameter tensor (reference: "
               "max of the element-wise and the norm error)", "tests_assert": "<= 10 x these, floor 2e-6"}
with open(out_path, "w") as f:
    json.dump(rec, f, indent=1, sort_keys=True)
print("wrote", out_path)
