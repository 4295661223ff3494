"""The flow-matching training step of the reflection LoRA (train_flux/train/model.py:164-238, `OminiModel.step`), with the
57-block stack on the HIP kernels and their backward (train/blocks.py).

What is PyTorch here, as in the inference path (north star: "timestep embed ... left in PyTorch-ROCm"): the timestep / guidance /
pooled-text embedding, the LoRA terms of the AdaLN linears and of x_embedder (a [1, D] x [D, r] product each), the final
AdaLayerNormContinuous + proj_out (64 output channels) and the MSE.  torch.autograd links those pieces and the block Functions;
every block re-computes itself in its backward (train_flux/flux/transformer.py:139-157, `gradient_checkpointing: true`).

Data parallelism ("DDP all-reduce of ~116 M LoRA grads"): ONE flat bf16 bucket over all LoRA gradients (232 MB at
r = 32: a single ring all-reduce is per-link bound on xGMI, so fewer, larger collectives).  With `configure_optimizers()` the factors
and their gradients LIVE in flat buffers (train/optim.py): the all-reduce runs on the gradient buffer as it lies and the averaging is
a scalar of the optimizer kernel; `allreduce_lora_grads` is the stand-alone form for callers that keep torch.optim (pack, reduce,
average, unpack).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from .. import engine as E
from .. import ops
from ..flux.modules import LoraLinear
from .blocks import BlockOpts, DoubleBlockFn, SingleBlockFn, double_weights, fused_lora, single_weights

BF = torch.bfloat16


def lora_parameters(transformer) -> List[torch.nn.Parameter]:
    """The trainable set of train/model.py:62-66,94-103: every LoRA factor, nothing else."""
    return [p for n, p in transformer.named_parameters() if "lora_" in n]


def _lora_term(lin, s: torch.Tensor) -> Optional[torch.Tensor]:
    """scaling * lora_B(lora_A(s)) of a LoraLinear as torch ops (autograd), or None."""
    if not isinstance(lin, LoraLinear):
        return None
    out = None
    for a in lin.active_adapters:
        t = F.linear(F.linear(s, lin.lora_A[a].weight), lin.lora_B[a].weight) * lin.scaling[a]
        out = t if out is None else out + t
    return out


class _RowLoraTerms(torch.autograd.Function):
    """out[i] = scaling * B_i (A_i s) for n sibling LoRA pairs and ONE input row s -- matrix-VECTOR products, written as broadcast
    multiplies and fp32 row sums with the intermediate t = A s rounded to the factors' dtype (as F.linear(s, A) rounds it).  As torch.bmm / `@` they
    went to hipBLASLt, whose batched K = 1 / N = 1 launches in the BACKWARD (dB = dout t^T, dt = B^T dout) held the host 8-11 ms each,
    twice a step, at the one point of a step where it has no lead over the GPU (the end of the backward: profiles/r05_train_step_gaps.md)."""

    @staticmethod
    def forward(ctx, A, B, s, scaling):                     # A [n, r, K], B [n, N, r], s [K] (no gradient: silu(temb) is an input here)
        t = (A.float() * s.float()).sum(-1).to(A.dtype)     # [n, r]
        out = ((B.float() * t.float()[:, None, :]).sum(-1) * scaling).to(B.dtype)    # [n, N]
        ctx.save_for_backward(A, B, s, t)
        ctx.scaling = scaling
        return out

    @staticmethod
    def backward(ctx, dout):
        A, B, s, t = ctx.saved_tensors
        g = dout.float() * ctx.scaling                      # [n, N]
        dB = (g[:, :, None] * t.float()[:, None, :]).to(B.dtype) if ctx.needs_input_grad[1] else None
        dA = None
        if ctx.needs_input_grad[0]:
            dt = (B.float() * g[:, :, None]).sum(1).to(A.dtype)  # [n, r]: the gradient of the rounded t, as autograd would carry it
            dA = (dt.float()[:, :, None] * s.float()).to(A.dtype)
        return dA, dB, None, None


def _lora_terms_batched(lins, s: torch.Tensor):
    """[_lora_term(l, s)[0] for l in lins] for sibling linears of ONE shape fed the same row s [1, K] -- the AdaLN linears of all
    blocks see the same silu(temb): one [n r, K] x [K] and one batched [N, r] x [r] matrix-vector product (_RowLoraTerms) instead of two
    tiny GEMMs per block (and four more in their backward).  Falls back to the per-linear form for mixed shapes / several adapters."""
    if not lins:
        return []
    ok = all(isinstance(l, LoraLinear) and len(l.active_adapters) == 1 for l in lins)
    if ok:
        a0 = lins[0].active_adapters[0]
        wa, wb = lins[0].lora_A[a0].weight, lins[0].lora_B[a0].weight
        ok = all(l.active_adapters[0] == a0 and l.lora_A[a0].weight.shape == wa.shape and l.lora_B[a0].weight.shape == wb.shape and
                 l.scaling[a0] == lins[0].scaling[a0] for l in lins)
    if not ok:
        return [None if (t := _lora_term(l, s)) is None else t[0] for l in lins]
    n, (r, K_) = len(lins), wa.shape
    A = torch.stack([l.lora_A[a0].weight for l in lins])                    # [n, r, K]
    B = torch.stack([l.lora_B[a0].weight for l in lins])                    # [n, N, r]
    out = _RowLoraTerms.apply(A, B, s.reshape(K_), float(lins[0].scaling[a0]))   # [n, N]
    return list(out.unbind(0))


class FluxTrainer:
    """Forward with a backward through the HIP blocks for ONE transformer; `model_config` as in config.yaml:5-8."""

    def __init__(self, transformer, model_config: Optional[dict] = None, gradient_checkpointing="auto", gradient_clip_val: Optional[float] = 0.5):
        """gradient_clip_val: the global L2 norm the gradients are clipped to before every optimizer update -- the reference's Trainer is
        built with `gradient_clip_val=training_config.get("gradient_clip_val", 0.5)` (train_flux/train/train.py:165; config.yaml does not
        override it), so 0.5 is the default here too; None or 0 = no clipping.
        gradient_checkpointing: True = the reference's training config (config.yaml `gradient_checkpointing: true`,
        transformer.py:139-157: every block re-runs its forward inside the backward); False = every block keeps its intermediates;
        "auto" (default) = keep them for as many blocks as fit the free HBM (~0.8 GB per block at 5632 tokens: all 57 blocks of one
        sample are ~45 GB of 288) and re-compute the rest.  The gradients are bit-identical in all three."""
        self.tr = transformer
        self.cfg = dict(model_config or {})
        if gradient_checkpointing not in (True, False, "auto"):
            raise ops.RFError(f"gradient_checkpointing must be True, False or 'auto', got {gradient_checkpointing!r}")
        self.gradient_checkpointing = gradient_checkpointing
        if gradient_clip_val is not None and gradient_clip_val < 0:
            raise ops.RFError(f"gradient_clip_val must be >= 0 or None, got {gradient_clip_val!r}")
        self.gradient_clip_val = gradient_clip_val or None
        self.kept_blocks = 0                                # blocks of the last forward that kept their intermediates (telemetry)
        if self.cfg.get("add_cond_attn", False) or not self.cfg.get("union_cond_attn", True):
            raise ops.RFError("the training path covers the shipped training config (union_cond_attn: true, add_cond_attn: false)")
        E.check_lora_placement(transformer)
        if getattr(transformer, "_rf_merged_lora", False):
            raise ops.RFError("training needs the LoRA factors as factors: call pipe.enable_merged_lora(False) first")
        bad = [n for n, m in transformer.named_modules() if isinstance(m, LoraLinear) and
               (n.endswith("norm1_context.linear") or n.endswith("norm_out.linear"))]
        if bad:
            raise ops.RFError(f"LoRA on {bad[:2]} is outside the FLUX-Corrector target list (config.yaml:53): not trainable here")
        for p in transformer.parameters():
            p.requires_grad_(False)                         # train/model.py:96
        for p in lora_parameters(transformer):
            p.requires_grad_(True)                          # :102-103
        self.eng = E.engine_for(transformer)
        self.latent_lora = bool(self.cfg.get("latent_lora", False))
        self.optimizer = None
        self._auto_fell_back = False                        # "auto" ran out of memory once: every block re-computes from then on
        self._keep_plan_frozen: Optional[List[bool]] = None
        self._last_keep_plan: List[bool] = []

    def configure_optimizers(self, optimizer_config: Optional[dict] = None, state_dtype: torch.dtype = BF):
        """train/model.py:94-117 (`configure_optimizers`): the optimizer over the LoRA factors, built from the reference's
        `optimizer_config` ({"type": "Prodigy" | "AdamW", "params": {...}}; default: the shipped config.yaml:55-61).  The factors move
        into ONE flat bucket (train/optim.py: gradients accumulate straight into it, the data-parallel all-reduce and the update run on
        it as it lies), so the engine's packed pointers are re-made."""
        from .optim import build_optimizer
        if optimizer_config is None:
            optimizer_config = {"type": "Prodigy", "params": {"lr": 1, "use_bias_correction": True, "safeguard_warmup": True, "weight_decay": 0.01}}
        self.optimizer = build_optimizer(lora_parameters(self.tr), optimizer_config, state_dtype=state_dtype)
        E.invalidate(self.tr)
        self.eng = E.engine_for(self.tr)
        return self.optimizer

    _PER_SAMPLE = ("x_0", "x_1", "t", "prompt_embeds", "pooled_prompt_embeds", "condition_latents")

    def training_step(self, batch: Dict[str, torch.Tensor], world_size: int = 1, group=None, generator: Optional[torch.Generator] = None,
                      sample_by_sample: Optional[bool] = None) -> torch.Tensor:
        """One optimisation step as the reference's Lightning loop runs it (zero_grad -> step() -> backward -> DDP all-reduce ->
        clip_grad_norm_(gradient_clip_val) -> optimizer.step): returns the detached loss.  Needs configure_optimizers() first.
        sample_by_sample (default: on for a batch of several samples unless gradient_checkpointing is True): forward + backward of one
        sample at a time, each with 1 / B of the loss -- the mean over the batch is the mean of the per-sample means, so the summed
        gradients are those of the batched loss up to the order of the bf16 accumulation, while only ONE sample's activations are alive
        (a sample of the 57-block model keeps ~45 GB; the reference's batch of 8, config.yaml:11, would not fit 288 GB at once).  The
        draws of t and x_1 are made for the whole batch first, exactly as step() makes them."""
        if self.optimizer is None:
            raise ops.RFError("FluxTrainer.training_step: call configure_optimizers() first")
        opt = self.optimizer
        if self.gradient_checkpointing == "auto" and not self._auto_fell_back:
            # "auto" sizes what it keeps from a per-block cost model against the free HBM; if the model is wrong for this geometry (or
            # another process took the memory) the step must not die: ONE retry with the reference's recompute for every block, which
            # then stays on for this trainer.  The draws are made before the attempt so that both attempts see the same batch.
            batch = dict(batch)
            with torch.no_grad():
                Bn0 = batch["x_0"].shape[0]
                if batch.get("t") is None:
                    batch["t"] = torch.sigmoid(torch.randn((Bn0,), device=batch["x_0"].device, generator=generator))
                if batch.get("x_1") is None:
                    batch["x_1"] = torch.randn(batch["x_0"].shape, device=batch["x_0"].device, dtype=batch["x_0"].dtype, generator=generator)
            try:
                return self._training_step(batch, world_size, group, generator, sample_by_sample)
            except torch.cuda.OutOfMemoryError:
                self._auto_fell_back = True
                torch.cuda.empty_cache()
        return self._training_step(batch, world_size, group, generator, sample_by_sample)

    def _training_step(self, batch, world_size, group, generator, sample_by_sample) -> torch.Tensor:
        loss = self._forward_backward(batch, generator, sample_by_sample)
        self._reduce_clip_update(world_size, group)
        return loss.detach()

    def _reduce_clip_update(self, world_size, group) -> None:
        opt = self.optimizer
        if world_size > 1:
            opt.bucket.all_reduce(world_size, group)
            opt.grad_scale = 1.0 / world_size
        if self.gradient_clip_val:
            opt.clip_grad_norm_(self.gradient_clip_val)      # norm of the averaged gradient, on the device
        opt.step()

    def capture_training_step(self, batch: Dict[str, torch.Tensor], world_size: int = 1, group=None, warmup: int = 2,
                              sample_by_sample: Optional[bool] = None):
        """The step as ONE hipGraph: zero_grad + forward + backward of `training_step` (~2700 launches at FLUX size, 230 ms of host
        enqueue per 270 ms step in eager mode -- GPU-bound, but one slow Python day from host-bound) are captured once and replayed;
        the tail -- gradient all-reduce (world_size > 1), clip, optimizer update: <= 8 launches whose scalar arguments change per step
        (AdamW's bias corrections) -- stays eager.  Returns `run(batch) -> loss`: the tensors of `batch` are copied into the captured
        step's static inputs (same shapes and the same optional keys as the example; without `t` / `x_1` the step draws them inside
        the graph from the default generator, which hipGraph replay advances), the graph is replayed, the tail runs.

        The warm-up steps run for real on a side stream (workspaces, per-stream scratch, the rope-table cache and, for
        gradient_checkpointing="auto", the keep-or-recompute plan are fixed there); parameters and optimizer state are restored
        afterwards, so capturing does not train.  Results are bit-equal to the eager step (tests/test_round6_gpu.py).  The step's
        activations live in the graph's private pool from then on (what the eager step peaks at)."""
        if self.optimizer is None:
            raise ops.RFError("FluxTrainer.capture_training_step: call configure_optimizers() first")
        opt = self.optimizer
        static = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        p0 = opt.bucket.param.detach().clone()
        sd0 = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in opt.state_dict().items()}

        def restore():
            with torch.no_grad():
                opt.bucket.param.copy_(p0)
            opt.load_state_dict(sd0)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._training_step(static, world_size, group, None, sample_by_sample)
            torch.cuda.current_stream().synchronize()
            restore()
            self._keep_plan_frozen = list(self._last_keep_plan) if self.gradient_checkpointing == "auto" else None
            # the warm-up's activations sit in the caching allocator's per-stream free lists (one set for the caller's stream if it ran
            # eager steps before, one for this side stream); the graph's private pool is a third: hand the first two back first
            torch.cuda.empty_cache()
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph, stream=side):
                    loss = self._forward_backward(static, None, sample_by_sample)
            finally:
                self._keep_plan_frozen = None
        torch.cuda.current_stream().wait_stream(side)
        tensor_keys = [k for k, v in static.items() if isinstance(v, torch.Tensor)]

        def run(new_batch: Dict[str, torch.Tensor]) -> torch.Tensor:
            if sorted(k for k, v in new_batch.items() if isinstance(v, torch.Tensor)) != sorted(tensor_keys):
                raise ops.RFError(f"captured training step: the batch must carry the tensors {sorted(tensor_keys)}")
            with torch.no_grad():
                for k in tensor_keys:
                    if new_batch[k] is not static[k]:
                        if new_batch[k].shape != static[k].shape:
                            raise ops.RFError(f"captured training step: {k} has shape {tuple(new_batch[k].shape)}, captured {tuple(static[k].shape)}")
                        static[k].copy_(new_batch[k])
            graph.replay()
            self._reduce_clip_update(world_size, group)
            return loss.detach()
        run.graph, run.static_batch = graph, static
        return run

    def _forward_backward(self, batch, generator, sample_by_sample) -> torch.Tensor:
        opt = self.optimizer
        opt.zero_grad()
        Bn = batch["x_0"].shape[0]
        if sample_by_sample is None:
            sample_by_sample = Bn > 1 and self.gradient_checkpointing is not True
        if sample_by_sample and Bn > 1:
            batch = dict(batch)
            with torch.no_grad():                                                            # the draws of step(), for the whole batch
                if batch.get("t") is None:
                    batch["t"] = torch.sigmoid(torch.randn((Bn,), device=batch["x_0"].device, generator=generator))
                if batch.get("x_1") is None:
                    batch["x_1"] = torch.randn(batch["x_0"].shape, device=batch["x_0"].device, dtype=batch["x_0"].dtype, generator=generator)
            loss = None
            for b in range(Bn):
                one = {k: (v[b:b + 1] if k in self._PER_SAMPLE and isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
                part = self.step(one) / Bn
                part.backward()
                loss = part.detach().float() if loss is None else loss + part.detach().float()
                dt = part.dtype
                del part
            loss = loss.to(dt)
        else:
            loss = self.step(batch, generator=generator)
            loss.backward()
        return loss.detach()

    # -------------------------------------------------------------------------------------------------- checkpoints
    def save_lora(self, path: str, adapter_name: Optional[str] = None) -> str:
        """train/model.py:87-92 (`OminiModel.save_lora`, called every `save_interval` steps by callbacks.py:68-74):
        `FluxPipeline.save_lora_weights(path, transformer_lora_layers=get_peft_model_state_dict(transformer), safe_serialization=True)`
        -> `<path>/pytorch_lora_weights.safetensors` with keys `transformer.<module>.lora_{A,B}.weight` (peft drops the adapter name
        from the key) -- the file `pipe.load_lora_weights(path)` and the search drivers' `lora_path` read.  Every factor is CLONED out
        of the flat bucket first: after configure_optimizers() all of them are views of one storage, which safetensors refuses (or, by
        version, splits) and which torch.save would serialise whole.  Returns the file name."""
        import os
        from safetensors.torch import save_file
        names = sorted({a for _n, m in self.tr.named_modules() if isinstance(m, LoraLinear) for a in m.lora_A})
        if adapter_name is None:
            if len(names) != 1:
                raise ops.RFError(f"save_lora: adapters {names} on the model: name the one to save")
            adapter_name = names[0]
        sd = {}
        for n, m in self.tr.named_modules():
            if isinstance(m, LoraLinear) and adapter_name in m.lora_A:
                sd[f"transformer.{n}.lora_A.weight"] = m.lora_A[adapter_name].weight.detach().clone().contiguous()
                sd[f"transformer.{n}.lora_B.weight"] = m.lora_B[adapter_name].weight.detach().clone().contiguous()
        if not sd:
            raise ops.RFError(f"save_lora: no LoRA factors of adapter {adapter_name!r} on the model")
        os.makedirs(path, exist_ok=True)
        fn = os.path.join(path, "pytorch_lora_weights.safetensors")
        save_file(sd, fn, metadata={"format": "pt"})
        return fn

    # -------------------------------------------------------------------------------------------------- pieces
    def _mods(self, temb: torch.Tensor, lora_on: bool):
        """Per-block modulation rows for one conditioning row temb [1, D]: the base AdaLN linears through the HIP GEMM (no grad),
        plus -- when LoRA is on for this token stream -- the LoRA terms of norm1.linear / norm.linear as autograd ops."""
        tr, D = self.tr, self.eng.D
        with torch.no_grad():
            table = self.eng.mod_table(temb, lora=False)[0]
            s = ops.silu(temb.to(BF).contiguous())
        nd = len(tr.transformer_blocks)
        dbl_img, dbl_txt, sgl = [], [], []
        # the LoRA terms of every block's AdaLN linear at once (same input row for all of them)
        lt_d = _lora_terms_batched([b.norm1.linear for b in tr.transformer_blocks], s) if lora_on else [None] * nd
        lt_s = _lora_terms_batched([b.norm.linear for b in tr.single_transformer_blocks], s) if lora_on else [None] * len(tr.single_transformer_blocks)
        for i in range(nd):
            m = table[i * 12 * D:i * 12 * D + 6 * D]
            dbl_img.append(m if lt_d[i] is None else m + lt_d[i])
            dbl_txt.append(table[i * 12 * D + 6 * D:(i + 1) * 12 * D])
        for j in range(len(tr.single_transformer_blocks)):
            m = table[nd * 12 * D + j * 3 * D:nd * 12 * D + (j + 1) * 3 * D]
            sgl.append(m if lt_s[j] is None else m + lt_s[j])
        out = table[-2 * D:]
        return dbl_img, dbl_txt, sgl, out

    def _embed(self, lin, x: torch.Tensor, lora_on: bool) -> torch.Tensor:
        base = E._base(lin)
        with torch.no_grad():
            y = ops.linear(x.contiguous(), base.weight, base.bias)
        lt = _lora_term(lin, x) if lora_on else None
        return y if lt is None else y + lt

    # -------------------------------------------------------------------------------------------------- forward
    def forward(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, guidance, txt_ids, img_ids,
                condition_latents=None, condition_ids=None, c_t: float = 0.0) -> torch.Tensor:
        """tranformer_forward (train_flux/flux/transformer.py:47-252) with autograd: hidden_states [B, S_img, in_ch] (= x_t),
        timestep in [0, 1] as the training step passes it (model.py:221).  Returns the prediction [B, S_img, in_ch]."""
        tr, eng, D = self.tr, self.eng, self.eng.D
        dtype = tr.dtype
        use_cond = condition_latents is not None
        with torch.no_grad():
            ts = timestep.to(dtype) * 1000                                                   # transformer.py:95-114
            gd = guidance.to(dtype) * 1000 if guidance is not None else None
            temb = eng.temb(ts, gd, pooled_projections.to(dtype))
            if use_cond:
                ct = torch.ones_like(ts) * c_t * 1000
                cg = torch.ones_like(gd) * 1000 if gd is not None else None
                cond_temb = eng.temb(ct, cg, pooled_projections.to(dtype))
            if txt_ids.ndim == 3:
                txt_ids = txt_ids[0]
            if img_ids.ndim == 3:
                img_ids = img_ids[0]
            cos, sin = eng.rope_tables(txt_ids, img_ids, condition_ids if use_cond else None)
        outs = []
        # activation budget of this forward ("auto"): what the device and torch's allocator have free now, minus a reserve for the
        # backward's own temporaries; a block's intermediates are ~24 [S, D] tensors + 2 [S, mlp] (double) / ~20 + 3 (single)
        S_all = hidden_states.shape[1] + encoder_hidden_states.shape[1] + (condition_latents.shape[1] if use_cond else 0)
        mlp = eng.mlp
        cost_d = 2 * S_all * (24 * D + 2 * mlp)
        cost_s = 2 * S_all * (20 * D + 3 * mlp)
        budget = 0
        frozen = self._keep_plan_frozen            # a captured step replays the decisions of its warm-up forward (no driver query in a capture)
        plan: List[bool] = []
        if frozen is None and self.gradient_checkpointing == "auto" and torch.is_grad_enabled() and not self._auto_fell_back:
            free = torch.cuda.mem_get_info(hidden_states.device)[0] + torch.cuda.memory_reserved(hidden_states.device) \
                - torch.cuda.memory_allocated(hidden_states.device)
            budget = int(0.85 * free) - (16 << 30)
        self.kept_blocks = 0

        def opts(cost):
            nonlocal budget
            if frozen is not None:
                keep = torch.is_grad_enabled() and frozen[len(plan)]
            else:
                keep = torch.is_grad_enabled() and (self.gradient_checkpointing is False or (self.gradient_checkpointing == "auto" and budget >= cost))
            plan.append(bool(keep))
            if keep:
                budget -= cost
                self.kept_blocks += 1
            return BlockOpts(self.latent_lora, not keep)
        for b in range(hidden_states.shape[0]):
            m_img, m_txt, m_sgl, m_out = self._mods(temb[b:b + 1], self.latent_lora)
            if use_cond:
                c_img, _, c_sgl, _ = self._mods(cond_temb[b:b + 1], True)
            x_img = self._embed(tr.x_embedder, hidden_states[b].to(dtype), self.latent_lora)
            x_cond = self._embed(tr.x_embedder, condition_latents[b].to(dtype), True) if use_cond else None
            with torch.no_grad():
                x_txt = ops.linear(encoder_hidden_states[b].to(dtype).contiguous(), tr.context_embedder.weight, tr.context_embedder.bias)
            St = x_txt.shape[0]
            for i, blk in enumerate(tr.transformer_blocks):
                a = blk.attn
                lo = (*fused_lora([a.to_q, a.to_k, a.to_v]), *fused_lora([a.to_out[0]]), *fused_lora([blk.ff.net[2]]))
                x_txt, x_img, x_cond = DoubleBlockFn.apply(double_weights(blk), opts(cost_d), x_txt, x_img, x_cond, m_txt[i], m_img[i],
                                                           c_img[i] if use_cond else None, cos, sin, *lo)
            x_main = torch.cat([x_txt, x_img], 0)
            for j, blk in enumerate(tr.single_transformer_blocks):
                a = blk.attn
                lo = (*fused_lora([a.to_q, a.to_k, a.to_v, blk.proj_mlp]), *fused_lora([blk.proj_out]))
                x_main, x_cond = SingleBlockFn.apply(single_weights(blk), opts(cost_s), x_main, x_cond, m_sgl[j],
                                                     c_sgl[j] if use_cond else None, cos, sin, *lo)
            x_img = x_main[St:]
            # norm_out (AdaLayerNormContinuous: scale FIRST) + proj_out, transformer.py:243-244 -- torch ops, autograd
            scale, shift = m_out[:D], m_out[D:]
            xn = (F.layer_norm(x_img.float(), (D,), eps=1e-6) * (1.0 + scale.float()) + shift.float()).to(dtype)
            outs.append(F.linear(xn, tr.proj_out.weight, tr.proj_out.bias))
        self._last_keep_plan = plan
        return torch.stack(outs, 0)

    # -------------------------------------------------------------------------------------------------- the step
    def step(self, batch: Dict[str, torch.Tensor], generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """train/model.py:164-238.  `batch`: x_0 [B, S_img, 64] + img_ids (the VAE-encoded target, encode_images), prompt_embeds,
        pooled_prompt_embeds, text_ids, condition_latents [B, S_c, 64] + condition_ids (position delta applied).  Optional `t` /
        `x_1` pin the draws (tests); otherwise t = sigmoid(randn), x_1 = randn as the reference.  Returns the loss (call .backward())."""
        x_0 = batch["x_0"]
        dev, dtype = x_0.device, self.tr.dtype
        Bn = x_0.shape[0]
        with torch.no_grad():
            t = batch.get("t")
            if t is None:
                t = torch.sigmoid(torch.randn((Bn,), device=dev, generator=generator))
            x_1 = batch.get("x_1")
            if x_1 is None:
                x_1 = torch.randn(x_0.shape, device=dev, dtype=x_0.dtype, generator=generator)
            t_ = t[:, None, None]
            x_t = ((1 - t_) * x_0 + t_ * x_1).to(dtype)                                      # :188
            guidance = torch.ones_like(t) if self.tr.config.guidance_embeds else None        # :209-213
            target = (x_1 - x_0)
        pred = self.forward(x_t, batch["prompt_embeds"], batch["pooled_prompt_embeds"], t, guidance, batch["text_ids"], batch["img_ids"],
                            batch.get("condition_latents"), batch.get("condition_ids"))
        return F.mse_loss(pred, target.to(pred.dtype), reduction="mean")                     # :235


# ------------------------------------------------------------------------------------------------------ data parallel
def allreduce_lora_grads(params: Sequence[torch.nn.Parameter], world_size: int, group=None) -> int:
    """Average the LoRA gradients over the data-parallel ranks with ONE all-reduce of a flat bucket (DDP's job in the reference's
    Lightning trainer).  Parameters without a gradient contribute zeros, so every rank reduces the same layout.  Returns the
    bucket size in bytes."""
    import torch.distributed as dist
    ps = [p for p in params if p.requires_grad]
    if not ps:
        return 0
    dt, dev = ps[0].dtype, ps[0].device
    flat = torch.zeros(sum(p.numel() for p in ps), dtype=dt, device=dev)
    off = 0
    for p in ps:
        if p.grad is not None:
            flat[off:off + p.numel()] = p.grad.reshape(-1)
        off += p.numel()
    if world_size > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= world_size
    off = 0
    for p in ps:
        g = flat[off:off + p.numel()].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += p.numel()
    return flat.numel() * flat.element_size()
