"""CPU tests of the host logic: CLI flag surface vs the reference, checkpoint formats, lr schedule,
C-ABI symbol export (no compute calls)."""
import ast
import importlib.util
import json
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "phased-consistency-model_amd")
REF = "/root/reference/code/text_to_image_sd15/train_pcm_lora_sd15.py"


def load_cli():
    spec = importlib.util.spec_from_file_location("pcm_cli", os.path.join(PKG, "train_pcm_lora_sd15.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_cli_accepts_reference_launch_line_with_abbreviation():
    cli = load_cli()
    a = cli.parse_args(["--pretrained_teacher_model=./stable-diffusion-v1-5", "--output_dir=out", "--tracker_project_nam=proj",
                        "--mixed_precision=fp16", "--resolution=512", "--lora_rank=64", "--learning_rate=5e-6", "--loss_type=huber",
                        "--adam_weight_decay=1e-3", "--max_train_steps=5000", "--max_train_samples=4000000",
                        "--dataloader_num_workers=16", "--w_min=4", "--w_max=5", "--validation_steps=500", "--checkpointing_steps=1000",
                        "--checkpoints_total_limit=10", "--train_batch_size=20", "--enable_xformers_memory_efficient_attention",
                        "--gradient_accumulation_steps=1", "--use_8bit_adam", "--report_to=wandb", "--resume_from_checkpoint=latest",
                        "--seed=453645634", "--num_ddim_timesteps=50", "--multiphase=4", "--gradient_checkpointing"])
    assert a.tracker_project_name == "proj" and a.multiphase == 4 and a.loss_type == "huber" and a.w_min == 4.0


@pytest.mark.skipif(not os.path.exists(REF), reason="/root/reference not mounted")
def test_cli_flags_and_defaults_match_reference():
    tree = ast.parse(open(REF).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "parse_args"][0]
    ref = {}
    for node in ast.walk(fn):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
            name = node.args[0].value.lstrip("-")
            kw = {k.arg: k.value for k in node.keywords}
            default = ast.literal_eval(kw["default"]) if "default" in kw else (False if "action" in kw else None)
            ref[name] = default
    cli = load_cli()
    ours = vars(cli.parse_args(["--pretrained_teacher_model", "x"]))
    assert len(ref) == 51
    for k, v in ref.items():
        assert k in ours, k
        if k != "pretrained_teacher_model":
            assert ours[k] == v, (k, ours[k], v)


def test_adv_cli_adds_adv_flags_with_reference_defaults():
    """train_pcm_lora_sd15_adv.py:741-742 adds --adv_weight (0.1) and --adv_lr (1e-5) to the same flag set."""
    sys.path.insert(0, PKG)
    spec = importlib.util.spec_from_file_location("pcm_cli_adv", os.path.join(PKG, "train_pcm_lora_sd15_adv.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    a = m.parse_args(["--pretrained_teacher_model", "x"])
    assert a.adv_weight == 0.1 and a.adv_lr == 1e-5 and a.multiphase == 8
    a = m.parse_args(["--pretrained_teacher_model=x", "--adv_weight=0.3", "--adv_lr", "3e-5", "--loss_type=huber"])
    assert a.adv_weight == 0.3 and a.adv_lr == 3e-5 and a.loss_type == "huber"
    ref = "/root/reference/code/text_to_image_sd15/train_pcm_lora_sd15_adv.py"
    if os.path.exists(ref):
        src = open(ref).read()
        assert re.search(r'"--adv_weight".{0,80}default=0\.1', src, re.S) and re.search(r'"--adv_lr".{0,80}default=1e-5', src, re.S)


def test_sdxl_cli_flag_surface_matches_reference():
    """train_pcm_lora_sdxl_adv.py: the SD1.5 flag set + 4 extra flags, with the SDXL script's own defaults (resolution 1024, w_min 3, multiphase 4)."""
    sys.path.insert(0, PKG)
    spec = importlib.util.spec_from_file_location("pcm_cli_sdxl", os.path.join(PKG, "train_pcm_lora_sdxl_adv.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    ours = vars(m.parse_args(["--pretrained_teacher_model", "x"]))
    ref_path = "/root/reference/code/text_to_image_sdxl/train_pcm_lora_sdxl_adv.py"
    if os.path.exists(ref_path):
        tree = ast.parse(open(ref_path).read())
        fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "parse_args"][0]
        ref = {}
        for node in ast.walk(fn):
            if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
                kw = {k.arg: k.value for k in node.keywords}
                ref[node.args[0].value.lstrip("-")] = ast.literal_eval(kw["default"]) if "default" in kw else (False if "action" in kw else None)
        assert len(ref) == 55
        for k, v in ref.items():
            assert k in ours, k
            if k != "pretrained_teacher_model":
                assert ours[k] == v, (k, ours[k], v)
    assert ours["resolution"] == 1024 and ours["multiphase"] == 4 and ours["adv_weight"] == 0.1


def test_capi_exports_every_declared_symbol():
    from pcm_amd import build as B
    from pcm_amd import capi
    header = open(os.path.join(ROOT, "include", "pcm_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(pcm_\w+)\s*\(", header, flags=re.M))
    declared |= set(re.findall(r"^\s*size_t\s+(pcm_\w+)\s*\(", header, flags=re.M))        # the *_workspace_bytes queries
    for variant in B.VARIANTS:              # bfloat16 / IEEE-half, product / tools: the same C ABI ...
        L = capi.Lib(B.build(variant=variant))
        for name in declared:
            assert hasattr(L.dll, name), f"{name} declared in include/pcm_hip.h but not exported by the {variant} build"
        # ... and the pcm_debug_* hooks (declared nowhere in the header) only in the TOOLS builds
        assert hasattr(L.dll, "pcm_debug_gemm_big_mode") == variant.startswith("tools"), variant
    queries = {n for n in declared if n.endswith("_workspace_bytes")} | {"pcm_last_error", "pcm_abi_version", "pcm_build_id", "pcm_act_dtype", "pcm_gemm_plan_code", "pcm_gemm_emits_chstats"}
    assert declared - queries == set(capi._PROTOS), (declared - queries) ^ set(capi._PROTOS)


def test_missing_library_fails_loudly(tmp_path):
    from pcm_amd import capi
    with pytest.raises(RuntimeError, match="no fallback"):
        capi.Lib(str(tmp_path / "nope.so"))


def test_checkpoint_formats_roundtrip(tmp_path):
    from emu_lib import emu_lib
    from oracle import pcm_math as M
    from pcm_amd import capi, checkpoint as ck
    from pcm_amd.model import LoraState
    from pcm_amd.unet_spec import UNetConfig
    from safetensors.torch import load_file
    capi.set_lib(emu_lib())
    try:
        cfg = UNetConfig(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, heads=2)
        lora = LoraState(cfg, 64, 8.0, "cpu", seed=3, b_std=0.01)
        ck.save_lora(lora, str(tmp_path))
        sd = load_file(str(tmp_path / "adapter_model.safetensors"))
        k0 = "base_model.model.down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.lora_A.weight"
        assert k0 in sd and sd[k0].shape == (64, 64)
        assert sd["base_model.model.down_blocks.0.resnets.0.conv1.lora_A.weight"].shape == (64, 64, 3, 3)
        assert sd["base_model.model.down_blocks.0.resnets.0.conv1.lora_B.weight"].shape == (64, 64, 1, 1)
        dl = load_file(str(tmp_path / "unet_lora" / "pytorch_lora_weights.safetensors"))
        assert set(dl) == {"unet." + k for k in sd}
        ko = load_file(str(tmp_path / "pcm_lora_kohya_converted.safetensors"))
        kk = "lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_out_0"
        assert kk + ".lora_down.weight" in ko and kk + ".lora_up.weight" in ko and float(ko[kk + ".alpha"]) == 8.0
        assert all(M.kohya_key(k) in ko for k in sd)              # oracle restatement of the reference's renaming rule
        cfgj = json.load(open(tmp_path / "adapter_config.json"))
        assert cfgj["r"] == 64 and cfgj["lora_alpha"] == 8.0 and "to_out.0" in cfgj["target_modules"]
        lora2 = LoraState(cfg, 64, 8.0, "cpu", seed=99)
        ck.load_lora(lora2, str(tmp_path))
        assert torch.equal(lora2.params, lora.params)
        # the three on-disk formats load back through the format-sniffing reader (kohya is fp16: compare at that precision)
        for name, tol in (("adapter_model.safetensors", 0.0), (os.path.join("unet_lora", "pytorch_lora_weights.safetensors"), 0.0),
                          ("pcm_lora_kohya_converted.safetensors", 2e-3)):
            l3 = ck.unet_lora_from_file(cfg, str(tmp_path / name), "cpu")
            assert l3.real_rank == 64 and l3.alpha == 8.0 and set(l3.modules) == set(lora.modules)
            assert float((l3.params - lora.params).abs().max()) <= tol * float(lora.params.abs().max()), name
    finally:
        capi.set_lib(None)


def test_lr_schedule_constant_ignores_warmup():
    cli = load_cli()
    a = cli.parse_args(["--pretrained_teacher_model", "x", "--learning_rate", "5e-6", "--lr_warmup_steps", "500"])
    assert cli.lr_at(a, 0) == 5e-6 and cli.lr_at(a, 10000) == 5e-6


def test_sd3_cli_flag_surface_matches_reference():
    """train_pcm_lora_sd3.py: every flag of the reference's SD3 parser with the reference's default, and run.sh's launch line parses."""
    sys.path.insert(0, PKG)
    spec = importlib.util.spec_from_file_location("pcm_cli_sd3", os.path.join(PKG, "train_pcm_lora_sd3.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    ours = vars(m.parse_args(["--pretrained_teacher_model", "x"]))
    ref_path = "/root/reference/code/text_to_image_sd3/train_pcm_lora_sd3.py"
    if os.path.exists(ref_path):
        tree = ast.parse(open(ref_path).read())
        fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "parse_args"][0]
        ref = {}
        for node in ast.walk(fn):
            if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
                kw = {k.arg: k.value for k in node.keywords}
                ref[node.args[0].value.lstrip("-")] = ast.literal_eval(kw["default"]) if "default" in kw else (False if "action" in kw else None)
        assert len(ref) >= 60
        for k, v in ref.items():
            assert k in ours, k
            if k != "pretrained_teacher_model":
                assert ours[k] == v, (k, ours[k], v)
    a = m.parse_args(["--pretrained_teacher_model=/x", "--output_dir=o", "--tracker_project_nam=p", "--mixed_precision=fp16", "--resolution=1024",
                      "--lora_rank=32", "--learning_rate=5e-6", "--loss_type=huber", "--adam_weight_decay=1e-3", "--max_train_steps=20000",
                      "--dataloader_num_workers=16", "--w_min=4", "--w_max=5", "--validation_steps=1000", "--checkpointing_steps=2000",
                      "--checkpoints_total_limit=10", "--train_batch_size=2", "--enable_xformers_memory_efficient_attention",
                      "--gradient_accumulation_steps=1", "--use_8bit_adam", "--resume_from_checkpoint=latest", "--seed=453645634",
                      "--report_to=wandb", "--num_euler_timesteps=100", "--multiphase=2"])
    assert a.lora_rank == 32 and a.num_euler_timesteps == 100 and a.multiphase == 2 and a.tracker_project_name == "p"


def test_sd3_checkpoint_format(tmp_path):
    """StableDiffusion3Pipeline.save_lora_weights layout at the REAL rank (the kernels' padding to 64 never reaches a file)."""
    from emu_lib import emu_lib
    from pcm_amd import capi, checkpoint as ck
    from pcm_amd.mmdit import sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig
    from safetensors.torch import load_file
    capi.set_lib(emu_lib())
    try:
        cfg = MMDiTConfig(sample_size=16, num_layers=2, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
                          pooled_projection_dim=64, pos_embed_max_size=12)
        lora = sd3_lora_state(cfg, 32, 8.0, "cpu", seed=3, b_std=0.01)
        ck.save_lora_sd3(lora, str(tmp_path))
        sd = load_file(str(tmp_path / "pytorch_lora_weights.safetensors"))
        assert len(sd) == 2 * (2 * 6 + 1)
        assert sd["transformer.transformer_blocks.0.attn.to_q.lora_A.weight"].shape == (32, 128)
        assert sd["transformer.transformer_blocks.1.ff.net.0.proj.lora_B.weight"].shape == (512, 32)
        assert sd["transformer.proj_out.lora_B.weight"].shape == (64, 32)
        assert not any("add_" in k or "context" in k for k in sd)
        lora2 = sd3_lora_state(cfg, 32, 8.0, "cpu", seed=99)
        ck.load_lora(lora2, str(tmp_path))
        assert torch.equal(lora2.params, lora.params)
    finally:
        capi.set_lib(None)


def test_sd3_adv_cli_flags_and_lora_lists():
    sys.path.insert(0, PKG)
    mods = {}
    for name in ("train_pcm_lora_sd3_adv", "train_pcm_lora_sd3_adv_stochastic"):
        spec = importlib.util.spec_from_file_location("pcm_cli_" + name, os.path.join(PKG, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[name] = m
    a = mods["train_pcm_lora_sd3_adv"].parse_args(["--pretrained_teacher_model=x", "--lora_rank=32", "--num_euler_timesteps=100", "--multiphase=2",
                                                    "--adv_weight=0.2", "--adv_lr", "2e-5", "--loss_type=huber"])
    assert a.adv_weight == 0.2 and a.adv_lr == 2e-5 and a.multiphase == 2 and a.loss_type == "huber"
    d = mods["train_pcm_lora_sd3_adv"].parse_args(["--pretrained_teacher_model=x"])
    assert d.adv_weight == 0.1 and d.adv_lr == 1e-5                                     # train_pcm_lora_sd3_adv.py:640-641
    adv = mods["train_pcm_lora_sd3_adv"]
    adv.STOCHASTIC = False
    det = adv.lora_targets()
    adv.STOCHASTIC = True
    sto = adv.lora_targets()
    adv.STOCHASTIC = False
    assert len(det) == 22 and "pos_embed.proj" in det and "pos_embed.proj" not in sto and len(sto) == 21
    ref_path = "/root/reference/code/text_to_image_sd3/train_pcm_lora_sd3_adv.py"
    if os.path.exists(ref_path):          # the list is the reference's, verbatim
        src = open(ref_path).read()
        blk = src[src.index("target_modules=[\n", src.index("transformer_lora_config = LoraConfig(")):]        # (a commented one-line list precedes it)
        blk = blk[:blk.index("]")]
        assert tuple(re.findall(r'"([^"]+)"', blk)) == det


def test_resume_restores_adam_step_and_moments(tmp_path):
    """checkpoint-N -> a fresh trainer -> the next optimizer step must equal the uninterrupted run (moments AND the bias-correction
    step, which lives in device memory for graph replay)."""
    from emu_lib import emu_lib
    from pcm_amd import capi, checkpoint as ck
    from pcm_amd.model import LoraState
    from pcm_amd.trainer import Distiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig
    capi.set_lib(emu_lib())
    try:
        cfg = UNetConfig(block_out_channels=(64, 64), layers_per_block=1, cross_attention_dim=64, heads=2)

        def bare(seed):
            lora = LoraState(cfg, 64, 8.0, "cpu", seed=seed, b_std=0.01)
            D = Distiller.__new__(Distiller)       # optimizer half only
            D.lora, D.cfg, D.world_size, D.pg, D.step_count, D.ema = lora, StepConfig(learning_rate=1e-3), 1, None, 0, None
            D.step_dev = torch.zeros(1, dtype=torch.int64)
            D.lr_dev = torch.full((1,), 1e-3)
            return D
        g = torch.Generator().manual_seed(0)
        grads = [torch.randn(bare(5).lora.numel, generator=g) * 1e-3 for _ in range(4)]
        A = bare(5)
        for i in range(3):
            A.lora.grads.copy_(grads[i]); A.optimizer_step()
        ck.save_state(A, str(tmp_path / "checkpoint-3"), 3)
        A.lora.grads.copy_(grads[3]); A.optimizer_step()
        Bd = bare(99)                                # different init: everything must come from the checkpoint
        assert ck.load_state(Bd, str(tmp_path / "checkpoint-3")) == 3
        assert Bd.step_count == 3 and int(Bd.step_dev) == 3
        Bd.lora.grads.copy_(grads[3]); Bd.optimizer_step()
        assert torch.allclose(Bd.lora.params, A.lora.params, rtol=0, atol=1e-7), float((Bd.lora.params - A.lora.params).abs().max())
    finally:
        capi.set_lib(None)


def test_lr_schedules_restarts_and_polynomial():
    """the two extra schedules the SD3 parser exposes knobs for (--lr_num_cycles, --lr_power), against the closed forms of diffusers' get_scheduler."""
    import math
    import types
    cli = load_cli()
    a = types.SimpleNamespace(learning_rate=1e-4, lr_warmup_steps=10, max_train_steps=110, lr_num_cycles=2, lr_power=2.0, lr_scheduler="cosine_with_restarts")
    assert cli.lr_at(a, 5) == 1e-4 * 0.5                                   # warm-up
    assert abs(cli.lr_at(a, 10) - 1e-4) < 1e-12                            # start of cycle 1
    assert abs(cli.lr_at(a, 35) - 1e-4 * 0.5 * (1 + math.cos(math.pi * 0.5))) < 1e-12
    assert abs(cli.lr_at(a, 60) - 1e-4) < 1e-12                            # hard restart at half of the decay span
    assert cli.lr_at(a, 110) == 0.0
    a.lr_scheduler = "polynomial"
    assert abs(cli.lr_at(a, 60) - ((1e-4 - 1e-7) * 0.5 ** 2 + 1e-7)) < 1e-15
    assert abs(cli.lr_at(a, 110) - 1e-7) < 1e-15 and cli.lr_at(a, 500) == 1e-7


def _load(name):
    sys.path.insert(0, PKG)
    spec = importlib.util.spec_from_file_location("pcm_cli_" + name, os.path.join(PKG, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_latent_shard_sources_rank_sharding_and_shapes(tmp_path):
    """--latents_dir shards of the three trainers: files are dealt round-robin to ranks, batches are drawn from the rank's own
    shards only, the unconditional embeddings come from the shards (SD1.5 / SD3) or are zeros (SDXL, :1216-1221)."""
    from safetensors.torch import save_file
    dev = torch.device("cpu")
    g = torch.Generator().manual_seed(0)
    # ---- SD1.5
    d15 = tmp_path / "sd15"
    d15.mkdir()
    for i in range(4):
        t = {"latents": torch.full((3, 4, 8, 8), float(i)), "prompt_embeds": torch.randn(3, 77, 768, generator=g)}
        if i == 1:
            t["uncond_prompt_embeds"] = torch.full((77, 768), 7.0)
        save_file(t, str(d15 / f"shard{i}.safetensors"))
    cli = load_cli()
    a = cli.parse_args(["--pretrained_teacher_model", "x", "--latents_dir", str(d15), "--train_batch_size", "5", "--seed", "3"])
    for rank in range(2):
        src = cli.LatentSource(a, rank, 2, dev)
        lat, pe = src.batch()
        assert lat.shape == (5, 4, 8, 8) and pe.shape == (5, 77, 768) and len(src) == 6 // 5
        assert set(lat[:, 0, 0, 0].tolist()) <= ({0.0, 2.0} if rank == 0 else {1.0, 3.0})        # shards rank::world
        assert src.uncond.shape == (5, 77, 768)
        assert bool((src.uncond == 7.0).all())                                                 # run-global: found in shard1, used by both ranks
    at = cli.parse_args(["--pretrained_teacher_model", "x", "--latents_dir", str(d15), "--train_batch_size", "2", "--max_train_samples", "4"])
    assert cli.LatentSource(at, 0, 2, dev).lat.shape[0] == 2                                   # 4 samples over 2 ranks
    with pytest.raises(SystemExit):
        cli.LatentSource(cli.parse_args(["--pretrained_teacher_model", "x"]), 0, 1, dev)       # neither shards nor --synthetic_data
    with pytest.raises(FileNotFoundError):
        cli.LatentSource(a, 5, 8, dev)                                                          # more ranks than shards
    s = cli.LatentSource(cli.parse_args(["--pretrained_teacher_model", "x", "--synthetic_data", "--resolution", "256", "--train_batch_size", "2"]), 0, 1, dev)
    assert s.batch()[0].shape == (2, 4, 32, 32)
    # caption dropout: a dropped caption's embedding IS the unconditional one
    ad = cli.parse_args(["--pretrained_teacher_model", "x", "--latents_dir", str(d15), "--train_batch_size", "64", "--proportion_empty_prompts", "0.5", "--seed", "3"])
    sdrop = cli.LatentSource(ad, 1, 2, dev)
    _, pe = sdrop.batch()
    frac = float((pe == 7.0).all(dim=(1, 2)).float().mean())
    assert 0.25 < frac < 0.75, frac
    with pytest.raises(ValueError):
        cli.LatentSource(cli.parse_args(["--pretrained_teacher_model", "x", "--synthetic_data", "--proportion_empty_prompts", "1.5"]), 0, 1, dev)
    # ---- SDXL
    dxl = tmp_path / "sdxl"
    dxl.mkdir()
    save_file({"latents": torch.randn(4, 4, 16, 16, generator=g), "prompt_embeds": torch.randn(4, 77, 2048, generator=g),
               "pooled_prompt_embeds": torch.randn(4, 1280, generator=g)}, str(dxl / "a.safetensors"))
    xl = _load("train_pcm_lora_sdxl_adv")
    ax = xl.parse_args(["--pretrained_teacher_model", "x", "--latents_dir", str(dxl), "--train_batch_size", "2", "--resolution", "128"])
    sx = xl.SdxlSource(ax, 0, 1, dev)
    lat, pe, pp = sx.batch()
    assert lat.shape == (2, 4, 16, 16) and pe.shape == (2, 77, 2048) and pp.shape == (2, 1280)
    assert float(sx.uncond.abs().max()) == 0.0 and float(sx.uncond_pooled.abs().max()) == 0.0 and sx.time_ids.tolist()[0] == [128, 128, 0, 0, 128, 128]
    # ---- SD3
    d3 = tmp_path / "sd3"
    d3.mkdir()
    save_file({"latents": torch.randn(4, 16, 8, 8, generator=g), "prompt_embeds": torch.randn(4, 20, 96, generator=g),
               "pooled_prompt_embeds": torch.randn(4, 64, generator=g), "uncond_prompt_embeds": torch.full((20, 96), 2.0),
               "uncond_pooled_prompt_embeds": torch.full((64,), 3.0)}, str(d3 / "a.safetensors"))
    s3 = _load("train_pcm_lora_sd3")
    from pcm_amd.mmdit_spec import MMDiTConfig
    mc = MMDiTConfig(sample_size=16, num_layers=1, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128, pooled_projection_dim=64,
                     pos_embed_max_size=12)
    a3 = s3.parse_args(["--pretrained_teacher_model", "x", "--latents_dir", str(d3), "--train_batch_size", "3"])
    src3 = s3.SD3Source(a3, 0, 1, dev, mc)
    lat, pe, pp = src3.batch()
    assert lat.shape == (3, 16, 8, 8) and pe.shape == (3, 20, 96) and pp.shape == (3, 64)
    assert src3.uncond.shape == (3, 20, 96) and bool((src3.uncond == 2.0).all()) and bool((src3.uncond_pooled == 3.0).all())
    syn = s3.SD3Source(s3.parse_args(["--pretrained_teacher_model", "x", "--synthetic_data", "--resolution", "128", "--train_batch_size", "2"]), 0, 1, dev, mc)
    lat, pe, pp = syn.batch()
    assert lat.shape == (2, 16, 16, 16) and pe.shape == (2, 154, 96) and pp.shape == (2, 64)


@pytest.mark.slow
def test_sd3_clis_end_to_end_on_the_emulator(tmp_path, monkeypatch):
    """train_pcm_lora_sd3.py (train, checkpoint, resume, final LoRA file), train_pcm_lora_sd3_adv.py (one D and one G step) and
    sample_pcm_lora_sd3.py (loads the trained LoRA) run as programs: PCM_CLI_DEVICE=cpu + the host emulator as the library."""
    from emu_lib import emu_lib
    from pcm_amd import capi
    from safetensors.torch import load_file, save_file
    g = torch.Generator().manual_seed(0)
    shards = tmp_path / "shards"
    shards.mkdir()
    save_file({"latents": torch.randn(6, 16, 8, 8, generator=g), "prompt_embeds": torch.randn(6, 5, 96, generator=g),
               "pooled_prompt_embeds": torch.randn(6, 64, generator=g), "uncond_prompt_embeds": torch.randn(5, 96, generator=g),
               "uncond_pooled_prompt_embeds": torch.randn(64, generator=g)}, str(shards / "a.safetensors"))
    monkeypatch.setenv("PCM_CLI_DEVICE", "cpu")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    capi.set_lib(emu_lib())
    from pcm_amd import precision
    precision.register_lib("fp16", emu_lib("f16"))      # --teacher_precision defaults to the reference's split: the teacher pass runs in the half build
    try:
        out = tmp_path / "out"
        common = ["--pretrained_teacher_model", "random", "--tiny_model", "--latents_dir", str(shards), "--train_batch_size", "2", "--lora_rank", "32",
                  "--learning_rate", "1e-3", "--num_euler_timesteps", "20", "--multiphase", "2", "--seed", "1", "--output_dir", str(out)]
        tr = _load("train_pcm_lora_sd3")
        tr.main(tr.parse_args(common + ["--max_train_steps", "3", "--checkpointing_steps", "2"]))
        assert (out / "checkpoint-2" / "optimizer.safetensors").exists() and (out / "checkpoint-2" / "pytorch_lora_weights.safetensors").exists()
        sd = load_file(str(out / "pytorch_lora_weights.safetensors"))
        assert sd["transformer.transformer_blocks.0.attn.to_q.lora_A.weight"].shape == (32, 128)
        assert float(sd["transformer.transformer_blocks.0.attn.to_q.lora_B.weight"].abs().max()) > 0          # B left zero init: it trained
        log = [json.loads(l) for l in open(out / "logs" / "text2image-fine-tune.jsonl")]
        assert [r["step"] for r in log] == [1, 2, 3] and all(r["loss"] > 0 for r in log)
        tr.main(tr.parse_args(common + ["--max_train_steps", "4", "--resume_from_checkpoint", "latest"]))           # resumes at 2, runs 3..4
        log = [json.loads(l) for l in open(out / "logs" / "text2image-fine-tune.jsonl")]
        assert [r["step"] for r in log] == [1, 2, 3, 3, 4]
        # sampler CLI with the trained adapter
        pe = tmp_path / "pe.safetensors"
        save_file({"prompt_embeds": torch.randn(2, 5, 96, generator=g), "pooled_prompt_embeds": torch.randn(2, 64, generator=g),
                   "uncond_prompt_embeds": torch.randn(5, 96, generator=g), "uncond_pooled_prompt_embeds": torch.randn(64, generator=g)}, str(pe))
        sm = _load("sample_pcm_lora_sd3")
        lat_path = tmp_path / "lat.safetensors"
        sm.main(sm.parse_args(["--pretrained_teacher_model", "random", "--tiny_model", "--lora_dir", str(out), "--prompt_embeds", str(pe),
                               "--num_inference_steps", "2", "--guidance_scale", "1.5", "--resolution", "64", "--output", str(lat_path)]))
        lat = load_file(str(lat_path))["latents"]
        assert lat.shape == (2, 16, 8, 8) and bool(torch.isfinite(lat).all())
        # adversarial trainer: one discriminator and one generator step
        adv = _load("train_pcm_lora_sd3_adv")
        out2 = tmp_path / "out_adv"
        adv.main(adv.parse_args([a if a != str(out) else str(out2) for a in common] + ["--max_train_steps", "2", "--loss_type", "huber"]))
        log = [json.loads(l) for l in open(out2 / "logs" / "text2image-fine-tune.jsonl")]
        assert "d_loss" in log[0] and "loss_cm" in log[1] and "g_loss" in log[1]
        sd = load_file(str(out2 / "pytorch_lora_weights.safetensors"))
        assert "transformer.pos_embed.proj.lora_A.weight" in sd and sd["transformer.pos_embed.proj.lora_A.weight"].shape == (32, 16, 2, 2)
        assert "transformer.transformer_blocks.0.norm1.linear.lora_B.weight" in sd
        # the sampler reads module set and rank from the file: the 22-entry adapter of the adversarial trainer loads as well
        sm.main(sm.parse_args(["--pretrained_teacher_model", "random", "--tiny_model", "--lora_file", str(out2 / "pytorch_lora_weights.safetensors"),
                               "--lora_scale", "1.0", "--prompt_embeds", str(pe), "--num_inference_steps", "1", "--resolution", "64", "--output", str(lat_path)]))
        from pcm_amd import checkpoint as ck
        from pcm_amd.mmdit_spec import MMDiTConfig
        tcfg = tr.model_config(tr.parse_args(["--pretrained_teacher_model", "random", "--tiny_model"]))
        lo = ck.sd3_lora_from_file(tcfg, str(out2 / "pytorch_lora_weights.safetensors"), "cpu", scale=4.0)
        assert lo.real_rank == 32 and "pos_embed.proj" in lo.modules and len(lo.modules) == len(sd) // 2
        m0 = lo.modules["transformer_blocks.0.attn.to_q"]
        assert torch.allclose(m0.A[:32], 2.0 * sd["transformer.transformer_blocks.0.attn.to_q.lora_A.weight"])       # sqrt(alpha) on every tensor
    finally:
        capi.set_lib(None)


@pytest.mark.slow
def test_unet_clis_end_to_end_on_the_emulator(tmp_path, monkeypatch):
    """train_pcm_lora_sd15.py, train_pcm_lora_sd15_adv.py and train_pcm_lora_sdxl_adv.py as programs on a narrow UNet
    (--tiny_model, PCM_CLI_DEVICE=cpu + the host emulator as the library): step loop, logs, checkpoints, final LoRA files."""
    from emu_lib import emu_lib
    from pcm_amd import capi
    from safetensors.torch import load_file, save_file
    g = torch.Generator().manual_seed(0)
    d15, dxl = tmp_path / "s15", tmp_path / "sxl"
    d15.mkdir(); dxl.mkdir()
    save_file({"latents": torch.randn(6, 4, 8, 8, generator=g), "prompt_embeds": torch.randn(6, 7, 64, generator=g),
               "uncond_prompt_embeds": torch.randn(7, 64, generator=g)}, str(d15 / "a.safetensors"))
    save_file({"latents": torch.randn(6, 4, 8, 8, generator=g), "prompt_embeds": torch.randn(6, 7, 64, generator=g),
               "pooled_prompt_embeds": torch.randn(6, 64, generator=g)}, str(dxl / "a.safetensors"))
    monkeypatch.setenv("PCM_CLI_DEVICE", "cpu")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    capi.set_lib(emu_lib())
    from pcm_amd import precision
    precision.register_lib("fp16", emu_lib("f16"))      # --teacher_precision defaults to the reference's split: the teacher pass runs in the half build
    try:
        def args(out, shards, extra):
            return ["--pretrained_teacher_model", "random", "--tiny_model", "--latents_dir", str(shards), "--train_batch_size", "1", "--learning_rate", "1e-3",
                    "--multiphase", "2", "--seed", "1", "--output_dir", str(out)] + extra
        import time as _t
        _t0 = _t.time()
        cli = load_cli()
        o1 = tmp_path / "o1"
        cli.main(cli.parse_args(args(o1, d15, ["--max_train_steps", "1", "--checkpointing_steps", "1", "--loss_type", "huber"])))
        assert (o1 / "checkpoint-1" / "trainer_state.json").exists()
        sd = load_file(str(o1 / "adapter_model.safetensors"))
        assert any(k.endswith("conv1.lora_A.weight") and v.shape[0] == 64 for k, v in sd.items())
        assert (o1 / "pcm_lora_kohya_converted.safetensors").exists() and (o1 / "unet_lora" / "pytorch_lora_weights.safetensors").exists()
        log = [json.loads(l) for l in open(o1 / "logs" / "text2image-fine-tune.jsonl")]
        assert [r["step"] for r in log] == [1] and all(r["loss"] > 0 and r["grad_norm"] > 0 for r in log)
        sm = _load("sample_pcm_lora_sd15")                      # sampler CLI with the adapter just trained
        sm.main(sm.parse_args(["--pretrained_teacher_model", "random", "--tiny_model", "--lora_dir", str(o1), "--synthetic_prompts", "1",
                               "--num_inference_steps", "2", "--guidance_scale", "2.0", "--resolution", "64", "--output", str(tmp_path / "l15.safetensors")]))
        assert load_file(str(tmp_path / "l15.safetensors"))["latents"].shape == (1, 4, 8, 8)
        smx = _load("sample_pcm_lora_sdxl")
        smx.main(smx.parse_args(["--pretrained_teacher_model", "random", "--tiny_model", "--synthetic_prompts", "1", "--num_inference_steps", "1",
                                 "--resolution", "64", "--output", str(tmp_path / "lxl.safetensors")]))
        assert bool(torch.isfinite(load_file(str(tmp_path / "lxl.safetensors"))["latents"]).all())
        print("sd15 base cli %.1f s" % (_t.time() - _t0)); _t0 = _t.time()
        adv = _load("train_pcm_lora_sd15_adv")
        o2 = tmp_path / "o2"
        adv.main(adv.parse_args(args(o2, d15, ["--max_train_steps", "2"])))
        log = [json.loads(l) for l in open(o2 / "logs" / "text2image-fine-tune.jsonl")]
        assert "d_loss" in log[0] and "loss_cm" in log[1]
        print("sd15 adv cli %.1f s" % (_t.time() - _t0)); _t0 = _t.time()
        xl = _load("train_pcm_lora_sdxl_adv")
        o3 = tmp_path / "o3"
        xl.main(xl.parse_args(args(o3, dxl, ["--max_train_steps", "1", "--resolution", "64", "--adv_weight", "0"])))   # (adv SDXL step: tests/test_emu_adv.py)
        log = [json.loads(l) for l in open(o3 / "logs" / "text2image-fine-tune.jsonl")]
        assert len(log) == 1 and (o3 / "adapter_model.safetensors").exists()
        print("sdxl adv cli %.1f s" % (_t.time() - _t0))
    finally:
        capi.set_lib(None)


def test_scale_lr_semantics_per_script():
    """--scale_lr multiplies lr by accumulation * batch * processes in the SD3 and SDXL scripts; the SD1.5 script defines the flag
    and never reads it (kept as is)."""
    s3 = _load("train_pcm_lora_sd3")
    a = s3.parse_args(["--pretrained_teacher_model", "x", "--scale_lr", "--learning_rate", "1e-6", "--train_batch_size", "4"])
    assert abs(s3.apply_scale_lr(a, 8) - 1e-6 * 1 * 4 * 8) < 1e-18 and abs(a.learning_rate - 3.2e-5) < 1e-18
    b = s3.parse_args(["--pretrained_teacher_model", "x", "--learning_rate", "1e-6"])
    assert s3.apply_scale_lr(b, 8) == 1e-6
    for ref, uses in (("/root/reference/code/text_to_image_sd15/train_pcm_lora_sd15.py", False),
                      ("/root/reference/code/text_to_image_sdxl/train_pcm_lora_sdxl_adv.py", True),
                      ("/root/reference/code/text_to_image_sd3/train_pcm_lora_sd3.py", True)):
        if os.path.exists(ref):
            assert ("if args.scale_lr" in open(ref).read()) == uses, ref
    import inspect
    xl = _load("train_pcm_lora_sdxl_adv")
    assert "args.scale_lr" in inspect.getsource(xl.main) and "scale_lr" not in inspect.getsource(load_cli().main)


def test_scheduler_position_follows_accelerate_per_script():
    """non-constant schedules: the SD1.5 / SDXL scripts' schedulers advance num_processes times per optimizer step (accelerate's
    AcceleratedScheduler on a schedule built with raw step counts); the SD3 scripts pre-multiply the counts, so theirs does not."""
    import inspect
    import types
    cli = load_cli()
    assert cli.sched_step(10, 1) == 10 and cli.sched_step(10, 8) == 80
    a = types.SimpleNamespace(learning_rate=1e-4, lr_warmup_steps=80, max_train_steps=1000, lr_scheduler="constant_with_warmup")
    # diffusers get_constant_schedule_with_warmup: lr_lambda(step) = step / max(1, warmup) (lr 0 at step 0, like the linear / cosine ramps)
    assert abs(cli.lr_at(a, cli.sched_step(9, 8)) - 1e-4 * 72 / 80) < 1e-12          # 8 GPUs: warm-up over after 10 optimizer steps
    assert cli.lr_at(a, 0) == 0.0
    assert cli.lr_at(a, cli.sched_step(10, 8)) == 1e-4
    assert "sched_step" in inspect.getsource(cli.main) and "sched_step" in inspect.getsource(_load("train_pcm_lora_sd15_adv").main)
    assert "sched_step" in inspect.getsource(_load("train_pcm_lora_sdxl_adv").main)
    assert "sched_step" not in inspect.getsource(_load("train_pcm_lora_sd3").main)
    for ref, mult in (("/root/reference/code/text_to_image_sd15/train_pcm_lora_sd15.py", False), ("/root/reference/code/text_to_image_sd3/train_pcm_lora_sd3.py", True)):
        if os.path.exists(ref):
            assert ("num_warmup_steps=args.lr_warmup_steps * accelerator.num_processes" in open(ref).read()) == mult


def test_build_staleness_is_keyed_on_content_not_on_file_times(tmp_path, monkeypatch):
    """round-5 review, item 9: build() decided by mtime, so a snapshot could run binaries older than its sources.  Now every object / library
    carries a content stamp (sha256 of its source, every header and its flags): an edit with the file time restored is stale, a touch is not;
    the library exports the identity of ALL its sources (pcm_build_id) and capi refuses one of the tree's own libraries with another id."""
    import os
    from pcm_amd import build as B
    from pcm_amd import capi
    src, hdr, out = tmp_path / "k.hip", tmp_path / "k.h", str(tmp_path / "k.o")
    src.write_text("int a;\n"); hdr.write_text("#define X 1\n")
    st = os.stat(src)
    stamp = B._digest([str(src), str(hdr)], ["-O3"])
    assert B._stale(out, stamp)                      # nothing built yet
    open(out, "w").write("obj"); B._mark(out, stamp)
    assert not B._stale(out, stamp)
    os.utime(src, (st.st_atime + 100, st.st_mtime + 100))                      # touched, same content: still current
    assert not B._stale(out, B._digest([str(src), str(hdr)], ["-O3"]))
    src.write_text("int b;\n"); os.utime(src, (st.st_atime, st.st_mtime))       # edited, file time restored: stale
    assert B._stale(out, B._digest([str(src), str(hdr)], ["-O3"]))
    hdr.write_text("#define X 2\n")
    assert B._digest([str(src), str(hdr)], ["-O3"]) != B._digest([str(src), str(hdr)], ["-O3", "-DPCM_TOOLS"])
    # the tree's own library names this tree's sources ...
    L = capi.Lib(capi.DEFAULT_LIB)
    assert L.build_id == B.source_id() + "-bf16" and L.dll.pcm_abi_version() >= 5
    # ... and one built from other sources is refused
    monkeypatch.setattr(B, "source_id", lambda: "0123456789abcdef")
    with pytest.raises(RuntimeError, match="built from other sources"):
        capi.Lib(capi.DEFAULT_LIB)
    monkeypatch.setenv("PCM_ALLOW_STALE_LIB", "1")
    capi.Lib(capi.DEFAULT_LIB)
