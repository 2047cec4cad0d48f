"""GPT-2 greedy generation harness around the attention op (BASELINE configs 1, 3, 4).

Counterpart -- not a copy -- of the reference's driver stack:
  model          reference gpt2.py:28-56 (GPT2), transformer_block.py:8-39, feed_forward.py:6-19, gelu.py:4-12
  configs        reference gpt2.py:58-73   (small / medium / large / xl)
  weight load    reference gpt2.py:16-25, :47-56 + safetensors_utils.py:6-24 (HF names, Conv1D weights transposed,
                 LM head tied to wte) -- from a LOCAL model.safetensors only (no network here)
  greedy loop    reference sampler.py:10-21 (full re-forward per token, no KV cache; stop at EOS or when the TOTAL
                 length reaches max_tokens; logits[0,-1,:] -> argmax -> append)
  timing         reference gpt2.py:93-134  (warm-up of prompt+3 tokens, then one timed run)

Differences: batch > 1 (independent prompts, config 4) with per-row EOS handling; prompts are token ids (tiktoken is
not installed offline; the reference prompt "Hello, I'm a language model," is supplied as ids); without --weights the
model has seeded random weights, which keeps the token-identity check (naive vs HIP backend) valid and harsh.

    python llm-inference_amd/gpt2_generation.py --backend CUSTOM_CUDA --model gpt2 --device cuda --max-tokens 512
    python -m torch.distributed.run --nproc-per-node N ... gpt2_generation.py --batch 64 --model gpt2-xl   (DP over prompts)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from dataclasses import dataclass

import torch
from torch import nn

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from causal_self_attention import CausalSelfAttention, ScaledDotProductAttentionBackend  # noqa: E402
from gelu import GELU  # noqa: E402

# "Hello, I'm a language model," under the GPT-2 BPE (reference gpt2.py:143).  From memory, NOT re-verified here:
# tiktoken is not installed in this image and would need the network for its BPE files.
REFERENCE_PROMPT_IDS = (15496, 11, 314, 1101, 257, 3303, 2746, 11)
EOS_TOKEN = 50256            # <|endoftext|>, reference tokenizer.py:30-31

GPT2_CONFIGS = {             # reference gpt2.py:59-71
    "gpt2":        dict(emb_dim=768,  n_heads=12, n_layers=12),
    "gpt2-medium": dict(emb_dim=1024, n_heads=16, n_layers=24),
    "gpt2-large":  dict(emb_dim=1280, n_heads=20, n_layers=36),
    "gpt2-xl":     dict(emb_dim=1600, n_heads=25, n_layers=48),
}
COMMON = dict(vocab_size=50257, context_length=1024, drop_rate=0.1, qkv_bias=True)



def _add_layer_norm_op():
    import causal_multihead_self_attention  # noqa: F401  (in-tree extension: registers the ops on import)

    return torch.ops.causal_multihead_self_attention.add_layer_norm


def fused_add_layer_norm(x: torch.Tensor, delta, ln: nn.LayerNorm):
    """(x + delta, ln(x + delta)) in ONE launch of gpt2_add_layernorm_f32_fwd (include/gpt2_layernorm.h); delta None:
    (x, ln(x)).  Device tensors only -- there is no CPU fallback."""
    if not x.is_cuda:
        raise RuntimeError("fused add + LayerNorm runs the HIP kernel only: the input must be a device tensor")
    return _add_layer_norm_op()(x, delta, ln.weight, ln.bias, ln.eps)


class Block(nn.Module):
    """Pre-LN block: x + attn(LN(x)); x + mlp(LN(x)) (transformer_block.py:26-39); dropouts are identity in eval()."""

    def __init__(self, cfg: dict, backend: ScaledDotProductAttentionBackend, fused_gelu: bool = False):
        super().__init__()
        d = cfg["emb_dim"]
        self.gelu = GELU(fused=fused_gelu)           # reference feed_forward.py:14; fused = one HIP launch instead of eight
        self.first_layer_norm = nn.LayerNorm(d)
        self.attention = CausalSelfAttention(d_in=d, d_out=d, max_context_length=cfg["context_length"],
                                             p_dropout=cfg["drop_rate"], num_heads=cfg["n_heads"],
                                             qkv_bias=cfg["qkv_bias"], scaled_dot_product_attention_backend=backend)
        self.second_layer_norm = nn.LayerNorm(d)
        self.mlp_in = nn.Linear(d, 4 * d)
        self.mlp_out = nn.Linear(4 * d, d)

    def forward(self, x):
        x = x + self.attention(self.first_layer_norm(x))
        return x + self.mlp_out(self.gelu(self.mlp_in(self.second_layer_norm(x))))

    def forward_split(self, x, delta):
        """The same block with every residual add fused into the LayerNorm that follows it.  In: the residual stream x
        and the output `delta` of the previous sublayer, not yet added (None at the first block).  Out: the same pair
        after this block: (x + delta + attention(...), mlp(...))."""
        x, h = fused_add_layer_norm(x, delta, self.first_layer_norm)
        x, h = fused_add_layer_norm(x, self.attention(h), self.second_layer_norm)
        return x, self.mlp_out(self.gelu(self.mlp_in(h)))


class GPT2Model(nn.Module):
    def __init__(self, cfg: dict, backend: ScaledDotProductAttentionBackend, last_position_only: bool = False,
                 fused_gelu: bool = False, fused_add_ln: bool = False, fused_step_ops: bool = False):
        super().__init__()
        self.cfg = cfg
        # opt-in: token + positional embedding as one HIP launch (include/gpt2_step.h); the graph loop then also picks
        # the next token with gpt2_greedy_step instead of seven torch launches
        self.fused_step_ops = fused_step_ops
        if fused_step_ops:
            _add_layer_norm_op()                          # (loads the extension: fail at construction if it is missing)
        self.last_position_only = last_position_only      # opt-in; default = reference behaviour
        self.fused_add_ln = fused_add_ln                  # opt-in: residual add + next LayerNorm as one HIP launch
        if fused_add_ln:
            _add_layer_norm_op()                          # fail at construction, loudly, if the extension is missing
        self.token_embeddings = nn.Embedding(cfg["vocab_size"], cfg["emb_dim"])
        self.positional_embeddings = nn.Embedding(cfg["context_length"], cfg["emb_dim"])
        self.blocks = nn.ModuleList(Block(cfg, backend, fused_gelu) for _ in range(cfg["n_layers"]))
        self.layer_norm = nn.LayerNorm(cfg["emb_dim"])
        self.output_layer = nn.Linear(cfg["emb_dim"], cfg["vocab_size"], bias=False)

    def embed(self, tokens: torch.Tensor) -> torch.Tensor:
        """wte[tokens] + wpe[arange(T)] (reference gpt2.py:43-44)."""
        if self.fused_step_ops:
            if not tokens.is_cuda:
                raise RuntimeError("fused_step_ops runs HIP kernels only: the tokens must be a device tensor")
            return torch.ops.causal_multihead_self_attention.embed_tokens(tokens, self.token_embeddings.weight,
                                                                          self.positional_embeddings.weight)
        n = tokens.shape[1]
        return self.token_embeddings(tokens) + self.positional_embeddings(torch.arange(n, device=tokens.device))

    def hidden_states(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens (B, T) -> residual stream (B, T, emb_dim) after the last block, before the final LayerNorm."""
        if self.fused_add_ln:
            x, delta = self.hidden_states_split(tokens)
            return x + delta
        x = self.embed(tokens)
        for blk in self.blocks:
            x = blk(x)
        return x

    def hidden_states_split(self, tokens: torch.Tensor):
        """fused_add_ln form: (x, delta) with the residual stream = x + delta -- the last MLP output is left for the final
        LayerNorm's launch to add (final_norm())."""
        x = self.embed(tokens)
        delta = None
        for blk in self.blocks:
            x, delta = blk.forward_split(x, delta)
        return x, delta

    def final_norm(self, x: torch.Tensor, delta) -> torch.Tensor:
        return fused_add_layer_norm(x, delta, self.layer_norm)[1]

    def forward(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens (B, T) -> logits (B, T, vocab) (reference gpt2.py:41-45), or (B, 1, vocab) with last_position_only."""
        if self.fused_add_ln:
            x, delta = self.hidden_states_split(tokens)
            if self.last_position_only:                   # LayerNorm is per position: normalise the one row that is used
                x, delta = x[:, -1:, :], delta[:, -1:, :]
            return self.output_layer(self.final_norm(x, delta))
        x = self.layer_norm(self.hidden_states(tokens))
        if self.last_position_only:
            x = x[:, -1:, :]
        return self.output_layer(x)

    @torch.no_grad()
    def load_hf_safetensors(self, path: str) -> None:
        """HF GPT-2 checkpoint names; Conv1D weights are stored (in, out) and transposed here; LM head = wte."""
        from safetensors import safe_open

        def put(param, t, transpose=False):
            t = t.T if transpose else t
            if param.shape != t.shape:
                raise ValueError(f"shape mismatch: parameter {tuple(param.shape)} vs checkpoint {tuple(t.shape)}")
            param.copy_(t)

        with safe_open(path, framework="pt") as f:
            g = lambda k: f.get_tensor(k) if k in f.keys() else f.get_tensor("transformer." + k)
            put(self.token_embeddings.weight, g("wte.weight"))
            put(self.positional_embeddings.weight, g("wpe.weight"))
            put(self.output_layer.weight, g("wte.weight"))
            put(self.layer_norm.weight, g("ln_f.weight")); put(self.layer_norm.bias, g("ln_f.bias"))
            for i, b in enumerate(self.blocks):
                p = f"h.{i}."
                put(b.first_layer_norm.weight, g(p + "ln_1.weight")); put(b.first_layer_norm.bias, g(p + "ln_1.bias"))
                put(b.second_layer_norm.weight, g(p + "ln_2.weight")); put(b.second_layer_norm.bias, g(p + "ln_2.bias"))
                put(b.attention.qkv_layer.weight, g(p + "attn.c_attn.weight"), True); put(b.attention.qkv_layer.bias, g(p + "attn.c_attn.bias"))
                put(b.attention.out_proj_layer.weight, g(p + "attn.c_proj.weight"), True); put(b.attention.out_proj_layer.bias, g(p + "attn.c_proj.bias"))
                put(b.mlp_in.weight, g(p + "mlp.c_fc.weight"), True); put(b.mlp_in.bias, g(p + "mlp.c_fc.bias"))
                put(b.mlp_out.weight, g(p + "mlp.c_proj.weight"), True); put(b.mlp_out.bias, g(p + "mlp.c_proj.bias"))


def build_model(name: str, backend: ScaledDotProductAttentionBackend, device, weights: str | None = None, seed: int = 0,
                last_position_only: bool = False, overrides: dict | None = None, fused_gelu: bool = False,
                init_on_device: bool = False, fused_add_ln: bool = False, fused_step_ops: bool = False) -> GPT2Model:
    """init_on_device: draw the random weights on `device` directly (seconds instead of tens of seconds for GPT2-XL;
    the values then come from the device's generator, not the CPU's -- fine for timing, and every rank that seeds alike
    gets the same weights).  The default draws them on the CPU, so that CPU and GPU models of one seed are equal."""
    cfg = {**COMMON, **GPT2_CONFIGS[name], **(overrides or {})}
    torch.manual_seed(seed)                       # seeded random weights unless a local checkpoint is given
    if init_on_device and torch.device(device).type != "cpu":
        with torch.device(device):
            model = GPT2Model(cfg, backend, last_position_only, fused_gelu, fused_add_ln, fused_step_ops)
    else:
        model = GPT2Model(cfg, backend, last_position_only, fused_gelu, fused_add_ln, fused_step_ops)
    if weights:
        model.load_hf_safetensors(weights)
    return model.eval().to(device)


@torch.no_grad()
def greedy_generate(model: nn.Module, prompt_ids, max_tokens: int, eos_token: int = EOS_TOKEN, device=None) -> list[list[int]]:
    """Greedy decoding with the reference's loop semantics (sampler.py:14-19), for one prompt or a batch of
    equal-length prompts.  A row that has produced EOS keeps being padded with EOS and ignored; the loop ends when
    every row has hit EOS or the total length reaches max_tokens.  Returns the token lists (EOS-padded rows trimmed
    after their first EOS)."""
    device = device or next(model.parameters()).device
    text = torch.as_tensor(prompt_ids, dtype=torch.long, device=device)
    if text.dim() == 1:
        text = text.unsqueeze(0)
    plen = text.shape[1]
    done = torch.zeros(text.shape[0], dtype=torch.bool, device=device)
    while text.shape[1] < max_tokens and not bool(done.all()):
        logits = model(text)                                       # (B, T or 1, vocab): full re-forward, no KV cache
        nxt = torch.argmax(logits[:, -1, :], dim=-1)               # first max wins, as torch.argmax in the reference
        nxt = torch.where(done, torch.full_like(nxt, eos_token), nxt)
        text = torch.cat((text, nxt.unsqueeze(1)), dim=1)
        done |= nxt == eos_token
    out = []
    for row in text.tolist():
        cut = len(row)
        for i in range(plen, len(row)):
            if row[i] == eos_token:
                cut = i + 1
                break
        out.append(row[:cut])
    return out


class BucketedGreedyGenerator:
    """Greedy decoding without a host round trip per token.

    The reference loop (sampler.py:14-19) re-runs the whole model on a sequence that grows by one token, reads the
    arg-max back with `.item()` and builds a new tensor: ~200 small launches and one sync per token, so a 512-token
    generation is launch-bound.  Here the tokens live in ONE static (B, max_tokens) device buffer and a step is a
    fixed sequence of device operations over the first L positions, L = the current length rounded up to `bucket`:

        hidden_states(tokens[:, :L]) -> row `pos` -> final LayerNorm -> LM head -> argmax -> tokens[:, pos+1]; pos += 1

    Every operation of the model is per-position except attention, which is causal, so the padded positions
    > pos cannot influence row `pos`: the step computes the reference's logits[0, -1, :] (only that row of the LM
    head).  With a fixed shape per bucket the step is captured once per bucket as a HIP graph
    (`use_graphs=True`, torch.cuda.CUDAGraph = hipGraph on ROCm) and replayed: max_tokens/bucket graphs in all, one
    launch per token.  `use_graphs=False` runs the identical step eagerly (CPU tests, debugging).

    EOS: a finished row keeps receiving EOS; the host looks at the `done` flags every `check_every` steps only,
    and rows are trimmed after their first generated EOS, so the returned tokens equal greedy_generate()'s."""

    def __init__(self, model: "GPT2Model", batch: int, max_tokens: int, bucket: int = 64, eos_token: int = EOS_TOKEN,
                 use_graphs: bool = True, check_every: int = 32, device=None, progress=None):
        self.model, self.batch, self.max_tokens, self.bucket, self.eos = model, batch, max_tokens, bucket, eos_token
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        self.use_graphs = use_graphs
        if use_graphs and self.device.type != "cuda":
            raise RuntimeError("HIP graphs need a GPU; pass use_graphs=False for the eager form of the same step")
        if max_tokens > model.cfg["context_length"]:
            raise ValueError("max_tokens exceeds the model's context length")
        self.check_every = check_every
        self.progress = progress                                       # callable(length) at every EOS check, or None
        self.cap = -(-max_tokens // bucket) * bucket                  # buffer length: whole buckets
        self.cap = min(self.cap, model.cfg["context_length"])
        self.tokens = torch.zeros((batch, self.cap), dtype=torch.long, device=self.device)
        self.pos = torch.zeros((1,), dtype=torch.long, device=self.device)     # index of the last valid token
        self.done = torch.zeros((batch,), dtype=torch.bool, device=self.device)
        self._graphs: dict[int, "torch.cuda.CUDAGraph"] = {}
        self._capture_stream = None
        self._pool = None

    def bucket_len(self, length: int) -> int:
        return min(-(-length // self.bucket) * self.bucket, self.cap)

    @torch.no_grad()
    def _step(self, L: int) -> None:
        m = self.model
        if getattr(m, "fused_add_ln", False):
            x, delta = m.hidden_states_split(self.tokens[:, :L])
            if x.shape[0] * x.shape[1] <= 1024:
                # LayerNorm is per position: normalising these few rows costs one launch, as one row would, and leaves ONE
                # tensor to pick the row from
                h = m.final_norm(x, delta).index_select(1, self.pos)
            else:
                h = m.final_norm(x.index_select(1, self.pos), delta.index_select(1, self.pos))
        else:
            x = m.hidden_states(self.tokens[:, :L])
            h = m.layer_norm(x.index_select(1, self.pos))             # (B, 1, emb_dim): LayerNorm is per position
        if getattr(m, "fused_step_ops", False):                       # arg-max, EOS rule and token write: one launch
            torch.ops.causal_multihead_self_attention.greedy_step(m.output_layer(h)[:, 0, :], self.done, self.tokens, self.pos, self.eos)
            self.pos.add_(1)
            return
        nxt = torch.argmax(m.output_layer(h)[:, 0, :], dim=-1)        # first max wins, as in the reference
        nxt = torch.where(self.done, torch.full_like(nxt, self.eos), nxt)
        self.done.logical_or_(nxt == self.eos)
        self.pos.add_(1)
        self.tokens.index_copy_(1, self.pos, nxt.unsqueeze(1))

    def _graph_for(self, L: int) -> "torch.cuda.CUDAGraph":
        g = self._graphs.get(L)
        if g is None:
            saved = (self.tokens.clone(), self.pos.clone(), self.done.clone())
            # Warm-up ON the capture stream (one stream for every bucket of this generator): lazy inits, GEMM workspace,
            # the kernel's LDS attribute -- and the attention op's per-stream workspace, which a capturing call can use
            # but not create, so that captured and eager steps run the same plans.
            if self._capture_stream is None:
                self._capture_stream = torch.cuda.Stream(self.device)
            side = self._capture_stream
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                self._step(L)
            torch.cuda.current_stream(self.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._pool, stream=side):   # one memory pool: the graphs never overlap
                self._step(L)
            if self._pool is None:
                self._pool = g.pool()
            for dst, src in zip((self.tokens, self.pos, self.done), saved):
                dst.copy_(src)                                        # the warm-up step advanced the state: undo
            self._graphs[L] = g
        return g

    @torch.no_grad()
    def generate(self, prompt_ids) -> list[list[int]]:
        prompt = torch.as_tensor(prompt_ids, dtype=torch.long, device=self.device)
        if prompt.dim() == 1:
            prompt = prompt.unsqueeze(0)
        if prompt.shape[0] != self.batch:
            raise ValueError(f"generator was built for batch {self.batch}, got {prompt.shape[0]} prompts")
        plen = prompt.shape[1]
        if plen < 1 or plen > self.max_tokens:
            raise ValueError("prompt length must be in [1, max_tokens]")
        self.tokens.zero_()
        self.tokens[:, :plen] = prompt
        self.pos.fill_(plen - 1)
        self.done.zero_()
        length = plen
        while length < self.max_tokens:
            L = self.bucket_len(length)
            if self.use_graphs:
                self._graph_for(L).replay()
            else:
                self._step(L)
            length += 1
            if (length - plen) % self.check_every == 0:
                all_done = bool(self.done.all())                       # the only host sync of the loop
                if self.progress is not None:
                    self.progress(length)
                if all_done:
                    break                                              # the reference stops at EOS (sampler.py:14)
        out = []
        for row in self.tokens[:, :length].tolist():
            cut = len(row)
            for i in range(plen, len(row)):
                if row[i] == self.eos:
                    cut = i + 1
                    break
            out.append(row[:cut])
        return out


class KVCacheGreedyGenerator:
    """Greedy decoding WITH a KV cache -- opt-in, SURVEY.md section 8(f) row 4.  The reference deliberately has none
    (README.md:12-13): its loop recomputes every earlier position for every token.  Here each layer keeps the keys and
    values of the earlier tokens in a static (B, max_tokens, d_model) pair; the prompt (all but its last token) is
    run once through the ordinary full-sequence path to fill them, and every further token costs ONE position:

        token[pos] -> embeddings -> per layer: LN, QKV GEMM (1 row), cmhsa_fp32_decode_step (appends k, v at row pos and
        returns softmax(q.K[0..pos]^T/8) V[0..pos]), out-proj, MLP -> final LN -> LM head -> argmax -> token[pos+1]

    `pos` lives in device memory and the kernel reads it there, so the step has a fixed shape and is captured as ONE
    HIP graph that is replayed for every token (use_graphs=True).  Mathematically the logits are those of the
    reference loop; numerically the GEMMs see one row instead of T, so the last bits differ and token identity with
    the cache-less loop is an empirical property of the margins (tests/test_generation.py checks it on the GPU).

    `decode_fn(qkv_new, k_cache, v_cache, pos_i32, num_heads) -> (B, d_model)` defaults to the HIP op; CPU tests inject
    a torch restatement (there is no CPU fallback in the product path)."""

    def __init__(self, model: "GPT2Model", batch: int, max_tokens: int, eos_token: int = EOS_TOKEN, use_graphs: bool = True,
                 check_every: int = 32, device=None, decode_fn=None, progress=None, few_row_linear: bool = False):
        self.model, self.batch, self.max_tokens, self.eos = model, batch, max_tokens, eos_token
        # opt-in: every nn.Linear of the step (one row per prompt) through the HIP few-row kernel (include/gpt2_linear.h),
        # bias and the MLP's GELU inside it, instead of torch's bias copy + fp32 GEMM tile kernel.  Up to 4 prompts: the
        # kernel is weight-streaming, each further row costs it ~15 % while the GEMM kernel's time stays flat (measured:
        # 2.6x faster than torch at 1 row, 1.4x at 4, even at 8 on GPT2-small's widths and slower on GPT2-XL's).  The LM
        # head (no bias to copy, already at HBM speed in hipBLASLt) goes through it at one prompt only.
        self.few_row_linear = few_row_linear
        if few_row_linear:
            import causal_multihead_self_attention  # noqa: F401

            if batch > 4:
                raise ValueError("few_row_linear is for at most 4 prompts (rows per linear layer); larger batches are GEMM-shaped")
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        if use_graphs and self.device.type != "cuda":
            raise RuntimeError("HIP graphs need a GPU; pass use_graphs=False for the eager form of the same step")
        if max_tokens > model.cfg["context_length"]:
            raise ValueError("max_tokens exceeds the model's context length")
        if decode_fn is None:
            import causal_multihead_self_attention  # noqa: F401  (registers the op; ImportError = no silent fallback)

            decode_fn = torch.ops.causal_multihead_self_attention.causal_multihead_self_attention_decode
        self.decode_fn, self.use_graphs, self.check_every, self.progress = decode_fn, use_graphs, check_every, progress
        d = model.cfg["emb_dim"]
        dev = self.device
        self.k_cache = [torch.zeros((batch, max_tokens, d), device=dev) for _ in model.blocks]
        self.v_cache = [torch.zeros((batch, max_tokens, d), device=dev) for _ in model.blocks]
        self.tokens = torch.zeros((batch, max_tokens + 1), dtype=torch.long, device=dev)    # +1: the last step's write slot
        self.pos64 = torch.zeros((1,), dtype=torch.long, device=dev)       # index of the token the step consumes
        self.pos32 = torch.zeros((1,), dtype=torch.int32, device=dev)      # the same, as the kernel wants it
        self.done = torch.zeros((batch,), dtype=torch.bool, device=dev)
        self._graph = None

    @torch.no_grad()
    def _prefill(self, n: int) -> None:
        """Run tokens[:, :n] through the blocks with the model's own full-sequence attention; keep every layer's K, V."""
        from causal_self_attention import scaled_dot_product_attention

        m = self.model
        x = m.token_embeddings(self.tokens[:, :n]) + m.positional_embeddings(torch.arange(n, device=self.device))
        for i, blk in enumerate(m.blocks):
            a = blk.attention
            qkv = a.qkv_layer(blk.first_layer_norm(x))
            q, k, v = (qkv[:, :, j * a.d_out:(j + 1) * a.d_out] for j in range(3))
            self.k_cache[i][:, :n] = k
            self.v_cache[i][:, :n] = v
            ctx = scaled_dot_product_attention(backend=a.scaled_dot_product_attention_backend, queries=q, keys=k, values=v,
                                               num_heads=a.num_heads, causal_mask=a.causal_mask)
            x = x + a.out_proj_layer(ctx)
            x = x + blk.mlp_out(blk.gelu(blk.mlp_in(blk.second_layer_norm(x))))

    def _linear(self, layer: nn.Linear, x: torch.Tensor, gelu: bool = False) -> torch.Tensor:
        return torch.ops.causal_multihead_self_attention.linear_rows(x, layer.weight, layer.bias, 1 if gelu else 0)

    def _ln_linear(self, x, delta, ln: nn.LayerNorm, layer: nn.Linear, gelu: bool = False):
        return torch.ops.causal_multihead_self_attention.ln_linear_rows(x, delta, ln.weight, ln.bias, ln.eps, layer.weight, layer.bias, 1 if gelu else 0)

    @torch.no_grad()
    def _step_fused(self) -> None:
        """The same step with the model's opt-in fusions: every residual add inside the LayerNorm behind it
        (fused_add_ln) and the next-token pick as one launch (fused_step_ops)."""
        m = self.model
        tok = self.tokens.index_select(1, self.pos64)
        x = m.token_embeddings(tok) + m.positional_embeddings(self.pos64)
        delta = None
        for i, blk in enumerate(m.blocks):
            a = blk.attention
            if self.few_row_linear:
                # residual add + LayerNorm inside the linear kernel behind it: five launches per layer
                x, qkv = self._ln_linear(x, delta, blk.first_layer_norm, a.qkv_layer)
                ctx = self.decode_fn(qkv[:, 0, :], self.k_cache[i], self.v_cache[i], self.pos32, a.num_heads)
                x, up = self._ln_linear(x, self._linear(a.out_proj_layer, ctx).unsqueeze(1), blk.second_layer_norm, blk.mlp_in, gelu=True)
                delta = self._linear(blk.mlp_out, up)
                continue
            x, h = fused_add_layer_norm(x, delta, blk.first_layer_norm)
            qkv = a.qkv_layer(h)
            ctx = self.decode_fn(qkv[:, 0, :], self.k_cache[i], self.v_cache[i], self.pos32, a.num_heads)
            x, h = fused_add_layer_norm(x, a.out_proj_layer(ctx).unsqueeze(1), blk.second_layer_norm)
            delta = blk.mlp_out(blk.gelu(blk.mlp_in(h)))
        hn = m.final_norm(x, delta)
        logits = (self._linear(m.output_layer, hn) if self.few_row_linear and self.batch == 1 else m.output_layer(hn))[:, 0, :]
        if getattr(m, "fused_step_ops", False):
            torch.ops.causal_multihead_self_attention.greedy_step(logits, self.done, self.tokens, self.pos64, self.eos)
            self.pos64.add_(1)
            self.pos32.add_(1)
            return
        nxt = torch.argmax(logits, dim=-1)
        nxt = torch.where(self.done, torch.full_like(nxt, self.eos), nxt)
        self.done.logical_or_(nxt == self.eos)
        self.pos64.add_(1)
        self.pos32.add_(1)
        self.tokens.index_copy_(1, self.pos64, nxt.unsqueeze(1))

    @torch.no_grad()
    def _step(self) -> None:
        m = self.model
        if getattr(m, "fused_add_ln", False):
            return self._step_fused()
        tok = self.tokens.index_select(1, self.pos64)                                  # (B, 1)
        x = m.token_embeddings(tok) + m.positional_embeddings(self.pos64)              # (B, 1, d)
        for i, blk in enumerate(m.blocks):
            a = blk.attention
            if self.few_row_linear:
                qkv = self._linear(a.qkv_layer, blk.first_layer_norm(x))
                ctx = self.decode_fn(qkv[:, 0, :], self.k_cache[i], self.v_cache[i], self.pos32, a.num_heads)
                x = x + self._linear(a.out_proj_layer, ctx).unsqueeze(1)
                x = x + self._linear(blk.mlp_out, self._linear(blk.mlp_in, blk.second_layer_norm(x), gelu=True))
                continue
            qkv = a.qkv_layer(blk.first_layer_norm(x))                                 # (B, 1, 3d)
            ctx = self.decode_fn(qkv[:, 0, :], self.k_cache[i], self.v_cache[i], self.pos32, a.num_heads)
            x = x + a.out_proj_layer(ctx).unsqueeze(1)
            x = x + blk.mlp_out(blk.gelu(blk.mlp_in(blk.second_layer_norm(x))))
        hn = m.layer_norm(x)
        nxt = torch.argmax((self._linear(m.output_layer, hn) if self.few_row_linear and self.batch == 1 else m.output_layer(hn))[:, 0, :], dim=-1)
        nxt = torch.where(self.done, torch.full_like(nxt, self.eos), nxt)
        self.done.logical_or_(nxt == self.eos)
        self.pos64.add_(1)
        self.pos32.add_(1)
        self.tokens.index_copy_(1, self.pos64, nxt.unsqueeze(1))

    def _capture(self) -> None:
        saved = [t.clone() for t in (self.tokens, self.pos64, self.pos32, self.done)]
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self._step()                                   # warm-up: lazy initialisations must not land in the capture
        torch.cuda.current_stream(self.device).wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, stream=side):   # captured on the stream that ran the warm-up
            self._step()
        for dst, src in zip((self.tokens, self.pos64, self.pos32, self.done), saved):
            dst.copy_(src)                                 # undo the warm-up step (its cache row is rewritten by the real one)

    @torch.no_grad()
    def generate(self, prompt_ids) -> list[list[int]]:
        prompt = torch.as_tensor(prompt_ids, dtype=torch.long, device=self.device)
        if prompt.dim() == 1:
            prompt = prompt.unsqueeze(0)
        if prompt.shape[0] != self.batch:
            raise ValueError(f"generator was built for batch {self.batch}, got {prompt.shape[0]} prompts")
        plen = prompt.shape[1]
        if plen < 1 or plen > self.max_tokens:
            raise ValueError("prompt length must be in [1, max_tokens]")
        self.tokens.zero_()
        self.tokens[:, :plen] = prompt
        self.done.zero_()
        length = plen
        if length < self.max_tokens:
            if plen > 1:
                self._prefill(plen - 1)                    # the last prompt token is the first decode step's input
            self.pos64.fill_(plen - 1)
            self.pos32.fill_(plen - 1)
            if self.use_graphs and self._graph is None:
                self._capture()
        while length < self.max_tokens:
            if self.use_graphs:
                self._graph.replay()
            else:
                self._step()
            length += 1
            if (length - plen) % self.check_every == 0:
                all_done = bool(self.done.all())           # the only host sync of the loop
                if self.progress is not None:
                    self.progress(length)
                if all_done:
                    break                                  # the reference stops at EOS (sampler.py:14)
        out = []
        for row in self.tokens[:, :length].tolist():
            cut = len(row)
            for i in range(plen, len(row)):
                if row[i] == self.eos:
                    cut = i + 1
                    break
            out.append(row[:cut])
        return out


@dataclass
class GenerationTiming:
    first_run_seconds: float
    second_run_seconds: float
    max_gpu_memory_allocated_bytes: int
    tokens: list


def time_greedy_generation(model, prompt_ids, max_tokens: int, device, generator=None) -> GenerationTiming:
    """Mirror of reference gpt2.py:104-134: warm-up run of prompt+3 tokens, then the timed run.  With a
    BucketedGreedyGenerator the warm-up is one full generation instead (it captures the graphs, the analogue of the
    reference's warm-up paying for lazy initialisation), and the timed run replays them."""
    plen = len(prompt_ids[0]) if isinstance(prompt_ids[0], (list, tuple)) else len(prompt_ids)
    sync = (lambda: torch.cuda.synchronize(device)) if torch.device(device).type == "cuda" else (lambda: None)
    run = (lambda n: generator.generate(prompt_ids)) if generator is not None else \
          (lambda n: greedy_generate(model, prompt_ids, n, device=device))
    t0 = time.perf_counter_ns()
    run(plen + 3)
    sync()
    first = (time.perf_counter_ns() - t0) / 1e9
    torch.manual_seed(0)
    if torch.device(device).type == "cuda":
        torch.cuda.reset_peak_memory_stats(device)
    t0 = time.perf_counter_ns()
    toks = run(max_tokens)
    sync()
    second = (time.perf_counter_ns() - t0) / 1e9
    peak = torch.cuda.max_memory_allocated(device) if torch.device(device).type == "cuda" else 0
    return GenerationTiming(first, second, peak, toks)


def shard_prompts(n_prompts: int, rank: int, world: int) -> range:
    """Data-parallel split of independent prompts: contiguous, sizes differ by at most one (config 4)."""
    base, extra = divmod(n_prompts, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="CUSTOM_CUDA", choices=[b.name for b in ScaledDotProductAttentionBackend])
    ap.add_argument("--model", default="gpt2", choices=list(GPT2_CONFIGS))
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--max-tokens", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1, help="number of copies of the reference prompt (all ranks together)")
    ap.add_argument("--weights", default=None, help="local HF model.safetensors; default: seeded random weights")
    ap.add_argument("--last-position-lm-head", action="store_true")
    ap.add_argument("--threads", type=int, default=0, help="CPU threads for --device cpu (0 = all cores)")
    ap.add_argument("--graphs", action="store_true", help="bucketed static-buffer loop replayed as HIP graphs (one launch per token)")
    ap.add_argument("--bucket", type=int, default=64)
    ap.add_argument("--few-row-linear", action="store_true", help="with --kv-cache: the step's linear layers through the HIP few-row kernel (bias and GELU inside)")
    ap.add_argument("--kv-cache", action="store_true", help="opt-in KV-cache decoding (one position per token, one HIP graph); not the reference's algorithm")
    ap.add_argument("--progress", action="store_true", help="with --graphs: a line on stderr at every EOS check (32 tokens)")
    ap.add_argument("--fused-gelu", action="store_true", help="GELU as one HIP launch (include/gpt2_gelu.h) instead of eight torch ops")
    ap.add_argument("--fused-step-ops", action="store_true", help="embedding sum and next-token pick (arg-max, EOS, token write) as one HIP launch each (include/gpt2_step.h)")
    ap.add_argument("--fused-add-ln", action="store_true", help="every residual add + the LayerNorm after it as one HIP launch (include/gpt2_layernorm.h)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = ScaledDotProductAttentionBackend[args.backend]
    if args.device == "cuda":
        if not torch.cuda.is_available():
            raise SystemExit("no GPU visible (the CUSTOM_CUDA backend has no CPU fallback)")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
        torch.set_num_threads(args.threads or (os.cpu_count() or 1))
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl" if device.type == "cuda" else "gloo",
                                **({"device_id": device} if device.type == "cuda" else {}))

    model = build_model(args.model, backend, device, weights=args.weights, last_position_only=args.last_position_lm_head,
                        fused_gelu=args.fused_gelu, fused_add_ln=args.fused_add_ln, fused_step_ops=args.fused_step_ops)
    mine = shard_prompts(args.batch, rank, world)
    prompts = [list(REFERENCE_PROMPT_IDS) for _ in mine]
    if world > 1:
        dist.barrier()
    progress = (lambda n: print(f"[rank {rank}] length {n}", file=sys.stderr, flush=True)) if args.progress else None
    gen = None
    if prompts and args.kv_cache:
        gen = KVCacheGreedyGenerator(model, len(prompts), args.max_tokens, device=device, progress=progress, few_row_linear=args.few_row_linear)
    elif prompts and args.graphs:
        gen = BucketedGreedyGenerator(model, len(prompts), args.max_tokens, bucket=args.bucket, device=device, progress=progress)
    timing = time_greedy_generation(model, prompts, args.max_tokens, device, gen) if prompts else GenerationTiming(0, 0, 0, [])
    seconds = timing.second_run_seconds
    generated = sum(len(t) - len(REFERENCE_PROMPT_IDS) for t in timing.tokens)
    if world > 1:
        agg = torch.tensor([seconds, float(generated)], dtype=torch.float64, device=device)
        mx = agg.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        seconds, generated = float(mx[0]), int(agg[1])
        gathered = [None] * world
        dist.all_gather_object(gathered, timing.tokens)          # token ids only: <= batch * max_tokens * 8 B
        tokens = [t for part in gathered for t in part]
        dist.destroy_process_group()
    else:
        tokens = timing.tokens
    if rank == 0:
        print(json.dumps({
            "model": args.model, "backend": args.backend, "loop": "kv cache, one hip graph" if args.kv_cache else f"hip graphs, bucket {args.bucket}" if args.graphs else "eager (reference loop)", "gelu": "fused HIP kernel" if args.fused_gelu else "reference expression (8 torch ops)", "add_layer_norm": "fused HIP kernel" if args.fused_add_ln else "torch add + nn.LayerNorm", "step_ops": "fused HIP kernels" if args.fused_step_ops else "torch ops", "device": str(device), "n_gpus": world if device.type == "cuda" else 0,
            "cpu_threads": torch.get_num_threads() if device.type == "cpu" else None,
            "weights": args.weights or "random (torch.manual_seed(0))", "prompts": args.batch, "max_tokens": args.max_tokens,
            "warmup_seconds": round(timing.first_run_seconds, 4), "seconds": round(seconds, 4),
            "tokens_generated": generated, "tokens_per_second": round(generated / seconds, 2) if seconds else None,
            "max_gpu_memory_allocated_kib": timing.max_gpu_memory_allocated_bytes // 1024,
            "first_tokens": tokens[0][:24] if tokens else [],
        }))


if __name__ == "__main__":
    main()
