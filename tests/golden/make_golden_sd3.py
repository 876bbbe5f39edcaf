"""Generate tests/golden/pcm_fm_golden.safetensors by EXECUTING THE REFERENCE'S OWN SOURCE (AST-sliced from
/root/reference/code/text_to_image_sd3, see oracle/ref_slice.py) on seeded inputs: the flow-matching PCM math of the SD3
trainer (EulerSolver, train_pcm_lora_sd3.py:158-230, and the step's expressions at :1285-1372) and the two PCM samplers
(pcm_fm_deterministic_scheduler.py / pcm_fm_stochastic_scheduler.py).

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_sd3.py
The fixture travels to the GPU box; /root/reference does not.
"""
import contextlib
import io
import os
import sys

import torch
from safetensors.torch import save_file

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import ref_slice  # noqa: E402


def main():
    assert ref_slice.available(), "/root/reference not mounted"
    ns = ref_slice.sd3_train_namespace()
    sig_table = ref_slice.sd3_flow_sigmas(1000, 3.0)
    out = {"flow_sigmas": torch.from_numpy(sig_table.copy())}
    solver = ns["EulerSolver"](sig_table, timesteps=1000, euler_timesteps=50)       # :961-965 (--num_euler_timesteps 50)
    out["euler_timesteps"] = solver.euler_timesteps
    out["euler_timesteps_prev"] = solver.euler_timesteps_prev
    out["sigmas"] = solver.sigmas                    # float32
    out["sigmas_prev"] = solver.sigmas_prev          # float64 (np.asarray of python floats, :168-170)
    g = torch.Generator().manual_seed(4321)
    B = 16
    shape = (B, 16, 6, 6)
    index = torch.cat([torch.tensor([0, 12, 13, 49, 25, 37, 24, 11]), torch.randint(0, 50, (B - 8,), generator=g)]).long()
    x = torch.randn(shape, generator=g)
    noise = torch.randn(shape, generator=g)
    pred = torch.randn(shape, generator=g)
    cond = torch.randn(shape, generator=g)
    uncond = torch.randn(shape, generator=g)
    out.update(index=index, x=x, noise=noise, pred=pred, cond=cond, uncond=uncond)
    ext = ns["extract_into_tensor"]
    # :1291-1301
    sigmas = ext(solver.sigmas, index, x.shape)
    sigmas_prev = ext(solver.sigmas_prev, index, x.shape)
    out["timesteps"] = (sigmas * 1000).squeeze()
    out["timesteps_prev"] = (sigmas_prev * 1000).squeeze()
    noisy = sigmas * noise + (1.0 - sigmas) * x
    out["noisy"] = noisy
    # :1313-1315 online branch, :1368-1370 target branch
    for M in (1, 2, 4, 8):
        xp, te = solver.euler_style_multiphase_pred(noisy, pred, index, M)
        out[f"online_{M}_x"], out[f"online_{M}_end"] = xp, te
        xp, te = solver.euler_style_multiphase_pred(noisy, pred, index, M, True)
        out[f"target_{M}_x"], out[f"target_{M}_end"] = xp, te
    # :1333-1357 teacher CFG (w = 3) + euler_step
    w = 3
    teacher = cond + w * (cond - uncond)
    out["teacher"] = teacher
    out["euler_step"] = solver.euler_step(noisy, teacher, index)
    # the step's own chain (:1357-1370): the target jump starts from the float64 x_prev of the Euler step
    for M in (1, 4):
        xp, te = solver.euler_style_multiphase_pred(out["euler_step"], pred, index, M, True)
        out[f"chain_target_{M}_x"], out[f"chain_target_{M}_end"] = xp, te
    # adversarial trainers' re-noising (train_pcm_lora_sd3_adv.py:1413-1445; the expression is inline in the training loop, so it is
    # evaluated here verbatim on the reference's own tables / extract_into_tensor rather than sliced as a function)
    end_index = out["online_4_end"]
    adv_index = end_index + torch.randint(0, 50 // 4, (B,), generator=g)
    adv_noise = torch.randn(shape, generator=g, dtype=torch.float64)
    sigmas_end = ext(solver.sigmas_prev, end_index, x.shape)
    sigmas_adv = ext(solver.sigmas_prev, adv_index, x.shape)
    out["adv_index"], out["adv_noise"] = adv_index, adv_noise
    out["timesteps_adv"] = (sigmas_adv * 1000).squeeze([1, 2, 3])
    out["fake_adv"] = ((1 - sigmas_adv) * out["online_4_x"] + (sigmas_adv - sigmas_end) * adv_noise) / (1 - sigmas_end)
    # :1374-1379 loss
    a, b = out["online_4_x"], out["target_4_x"]
    out["huber_loss"] = torch.mean(torch.sqrt((a.float() - b.float()) ** 2 + 0.001 ** 2) - 0.001).reshape(1)

    # ---- samplers (validation pipeline uses PCMFMDeterministicScheduler(1000, 3.0, 100), :1453) ----
    for kind in ("deterministic", "stochastic"):
        cls = ref_slice.sd3_sampler_class(kind)
        for steps in (1, 2, 4, 8):
            sch = cls(1000, 3.0, 100)
            with contextlib.redirect_stdout(io.StringIO()):      # set_timesteps prints its table
                sch.set_timesteps(steps)
            if kind == "deterministic":
                out[f"sampler_timesteps_{steps}"] = sch.timesteps.clone()
                out[f"sampler_sigmas_{steps}"] = sch.sigmas_.clone()
            g2 = torch.Generator().manual_seed(99 + steps)
            lat = torch.randn(2, 16, 6, 6, generator=g2)
            out[f"{kind}_{steps}_x0"] = lat.clone()
            for i, t in enumerate(sch.timesteps):
                v = torch.randn(2, 16, 6, 6, generator=g2)
                out[f"{kind}_{steps}_v{i}"] = v
                if kind == "stochastic":
                    torch.manual_seed(1000 * steps + i)
                    out[f"{kind}_{steps}_noise{i}"] = torch.randn_like(lat)      # what the step's randn_like will draw
                    torch.manual_seed(1000 * steps + i)
                lat = sch.step(v, t, lat, return_dict=False)[0]
                out[f"{kind}_{steps}_x{i + 1}"] = lat.clone()
    path = os.path.join(os.path.dirname(__file__), "pcm_fm_golden.safetensors")
    save_file({k: v.contiguous() for k, v in out.items()}, path)
    print("wrote", path, len(out), "tensors", os.path.getsize(path), "bytes")
    for k in ("sigmas", "sigmas_prev", "noisy", "online_4_x", "target_4_x", "euler_step", "timesteps_prev"):
        print(k, out[k].dtype, tuple(out[k].shape))


if __name__ == "__main__":
    main()
