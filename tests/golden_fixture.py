"""TEST INFRASTRUCTURE: committed oracle outputs for the GPU parity tests (tests/golden/step_*.safetensors).

The CPU oracle is deterministic on seeded CPU weights and inputs, so its outputs are computed ONCE in the build container by
tests/golden/make_golden_step.py (which runs the ``ref_*`` builders of tests/step_golden_cases.py) and committed; the ``-m gpu`` tests load
them instead of re-evaluating minutes of CPU oracle on the GPU box.  A fixture also pins the oracle against host differences (torch's CPU
sqrt / BLAS summation order differ between hosts in the last bit).

Large vectors (67 M LoRA gradients, 664 M discriminator-head gradients) are stored as a linear COUNT-SKETCH of 2^16 buckets:
    S(v)[h(i)] += s(i) * v[i],      h, s = fixed hashes of the element index.
S is linear, so |S(a) - S(b)|^2, |S(b)|^2 and <S(a), S(b)> are unbiased estimates of |a - b|^2, |b|^2 and <a, b> with relative standard
deviation sqrt(2 / 2^16) = 0.55 % -- rel-L2, cosine and norm ratios of the full vectors are read off 0.5 MB fixtures."""
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
M_SKETCH = 1 << 16
_P = (1 << 31) - 1                     # Mersenne prime; indices stay below 2^30, so (i * A + B) fits in int64 without wrap-around
_A1, _B1, _A2, _B2 = 1103515245, 12345, 1664525, 1013904223


def sketch(t, m=M_SKETCH, offset=0, out=None):
    """count-sketch of the flattened tensor (float64 [m], on the CPU); element order = the tensor's own (row-major) order, element i of
    ``t`` counted as global index ``offset + i``.  Evaluated on the tensor's device (fp64 sums: the result does not depend on it)."""
    flat = t.detach().reshape(-1)
    n = flat.numel()
    assert offset + n < (1 << 30)
    acc = torch.zeros(m, dtype=torch.float64, device=flat.device)
    CH = 1 << 24
    for s in range(0, n, CH):
        e = min(n, s + CH)
        i = torch.arange(offset + s, offset + e, dtype=torch.int64, device=flat.device)
        h = ((i * _A1 + _B1) % _P) % m
        sg = (((i * _A2 + _B2) % _P) & 1).to(torch.float64) * 2 - 1
        acc.index_add_(0, h, flat[s:e].to(torch.float64) * sg)
    acc = acc.cpu()
    return acc if out is None else out.add_(acc)


def sketch_cat(tensors, m=M_SKETCH):
    """sketch of torch.cat([t.reshape(-1) for t in tensors]), tensor by tensor (no concatenation, each on its own device)"""
    out, off = torch.zeros(m, dtype=torch.float64), 0
    for t in tensors:
        sketch(t, m, offset=off, out=out)
        off += t.numel()
    return out


def sk_rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def sk_cos(a, b):
    a, b = a.double(), b.double()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


def path(name):
    return os.path.join(GOLDEN, "step_%s.safetensors" % name)


def source_stamp():
    """what a fixture was computed FROM: sha256 over the oracle sources and the builders, plus the torch version.  Stored in the
    fixture's metadata when it is written; ``load`` warns when a stamped fixture no longer matches the tree (a later edit of oracle/*.py
    or of the seeded input builders would otherwise leave the GPU tests comparing against stale numbers without anyone noticing)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(HERE)
    for f in sorted(glob.glob(os.path.join(root, "oracle", "*.py"))) + [os.path.join(HERE, n) for n in ("step_golden_cases.py", "adv_cases.py")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return {"oracle_sha256": h.hexdigest(), "torch": torch.__version__}


def save(name, d):
    """d: {key: tensor | float | int | list of floats}.  Scalars / lists go to the metadata JSON (exact repr)."""
    from safetensors.torch import save_file
    os.makedirs(GOLDEN, exist_ok=True)
    tensors, meta = {}, {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            tensors[k] = v.detach().cpu().contiguous()
        else:
            meta[k] = v
    if not tensors:
        tensors["_empty"] = torch.zeros(1)
    save_file(tensors, path(name), metadata={"scalars": json.dumps(meta), "stamp": json.dumps(source_stamp())})


def stamp_of(name):
    from safetensors import safe_open
    with safe_open(path(name), framework="pt") as f:
        s = (f.metadata() or {}).get("stamp")
    return json.loads(s) if s else None


def load(name):
    from safetensors import safe_open
    out = {}
    with safe_open(path(name), framework="pt") as f:
        for k in f.keys():
            if k != "_empty":
                out[k] = f.get_tensor(k)
        md = f.metadata() or {}
        out.update(json.loads(md.get("scalars", "{}")))
    if md.get("stamp"):      # (fixtures written before round 5 carry no stamp)
        st, now = json.loads(md["stamp"]), source_stamp()
        if st.get("oracle_sha256") != now["oracle_sha256"]:
            import warnings
            warnings.warn("oracle fixture %s was written from other oracle / builder sources than this tree's (stamp %s..., tree %s...): "
                          "re-run tests/golden/make_golden_step.py --only %s" % (name, st.get("oracle_sha256", "")[:10], now["oracle_sha256"][:10], name))
    return out


def golden(name, builder):
    """The oracle side of a parity test.  PCM_GOLDEN_WRITE=1: evaluate ``builder()`` (minutes of CPU) and write the fixture;
    PCM_LIVE_ORACLE=1: evaluate it without touching the file (what the emulator suite's narrow cases always do -- they pass name=None);
    otherwise load the committed fixture and fail loudly if it is missing."""
    if name is None or os.environ.get("PCM_LIVE_ORACLE") == "1":
        return builder()
    if os.environ.get("PCM_GOLDEN_WRITE") == "1":
        d = builder()
        save(name, d)
        return load(name)
    if not os.path.exists(path(name)):
        raise FileNotFoundError("oracle fixture %s is missing: run `python tests/golden/make_golden_step.py --only %s` in the build container"
                                % (path(name), name))
    return load(name)
