"""Text conditioning.  The CLIP ViT-L/14 text encoder runs ONCE per sequence (`get_learned_conditioning([""])`,
test_any_image_pose.py:196-198), far off the hot path.  On a GPU its arithmetic runs on this library's kernels (round 4,
`ClipTextEngine`: LayerNorm folded into the q|k|v and fc1 GEMMs, causal md_attention, quick-GELU as SiLU of pre-scaled weights, the
two-term fp16 residual stream); the `transformers` module stays the PARAMETER CONTAINER (checkpoint keys
``cond_stage_model.transformer.*``) and the arithmetic of CPU-only use (tests, tokenizer checks) and of the non-default
``layer`` choices.  This module is the YAML ``cond_stage_config`` target with the reference's constructor kwargs, attribute names
(``tokenizer`` / ``transformer`` -> checkpoint keys ``cond_stage_model.transformer.*``) and ``encode`` / ``forward`` behaviour
(model_lib/ControlNet/ldm/modules/encoders/modules.py:88-131), made to work WITHOUT network access:

  * architecture: when ``from_pretrained(version)`` cannot reach a Hugging Face cache, the text tower is built from the
    ViT-L/14 text configuration embedded below (weights then come from the MagicDance checkpoint's ``cond_stage_model.*`` keys
    through ``load_state_dict``; magicdance_amd.cldm.adapt_clip_keys bridges the transformers 4.x <-> 5.x key layouts);
  * tokenizer: when the BPE vocabulary is absent, a fallback tokenizer serves the ONE prompt the entry points use by default, the
    empty string -- ``[BOS] [EOS] [EOS]...`` (the CLIP tokenizer pads with its EOS token, id 49407; BOS is 49406) -- and raises for
    anything else (a real prompt needs the vocabulary files, or pass --context_embedding);
  * the embedding of the empty prompt is cached: every frame batch and the unconditional branch ask for the same [1,77,768].
"""
import torch
import torch.nn as nn

# openai/clip-vit-large-patch14 text tower (config.json of the model card; also what SD-1.5 checkpoints carry)
VIT_L14_TEXT_CONFIG = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, projection_dim=768, num_hidden_layers=12,
                           num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                           attention_dropout=0.0, initializer_range=0.02, initializer_factor=1.0, pad_token_id=1,
                           bos_token_id=49406, eos_token_id=49407)
BOS_ID, EOS_ID = 49406, 49407


class EmptyPromptTokenizer:
    """Stand-in for CLIPTokenizer when its vocabulary files are not available: exact for "" (and only for "")."""

    def __init__(self, max_length=77):
        self.model_max_length = max_length

    def __call__(self, text, truncation=True, max_length=77, return_length=True, return_overflowing_tokens=False,
                 padding="max_length", return_tensors="pt"):
        text = [text] if isinstance(text, str) else list(text)
        if any(t != "" for t in text):
            raise RuntimeError("the CLIP BPE vocabulary (vocab.json / merges.txt of openai/clip-vit-large-patch14) is not available "
                               "offline: only the empty prompt can be tokenized -- provide the tokenizer files in the Hugging Face "
                               "cache, or pass the text context as a tensor (--context_embedding)")
        ids = torch.full((len(text), max_length), EOS_ID, dtype=torch.long)
        ids[:, 0] = BOS_ID
        return {"input_ids": ids, "length": torch.full((len(text),), 2, dtype=torch.long)}


class ClipTextEngine:
    """The CLIP text tower (transformers CLIPTextModel: embeddings -> 12 x [LN1, causal self-attention, LN2, quick-GELU MLP] ->
    final LayerNorm; the arithmetic behind encoders/modules.py:118-131 `self.transformer(input_ids=...)`) as C-ABI launches:

      x            = token_embedding[ids] + position_embedding                                (gather + md_add_f16)
      q|k , V^T    = md_igemm(x, LN1 folded into [3c][c] weights, q columns * d^-0.5 log2 e)   (one launch, V written transposed)
      a            = md_attention(q, k, V^T, causal)                                           (12 heads, d = 64, 77 tokens)
      x            = md_igemm(a, out_proj) + x                                                (two-term fp16 residual stream)
      h            = SiLU(md_igemm(x, LN2 folded into 1.702 * fc1))   = 1.702 * quick_gelu(fc1(LN2(x)))
      x            = md_igemm(h, fc2 / 1.702) + x
      out          = md_layernorm(x, final_layer_norm)

    quick_gelu(y) = y * sigmoid(1.702 y) = SiLU(1.702 y) / 1.702, so the activation the kernels already have serves with the factor
    moved into the fp32 master weights before the fp16 rounding.  Weights are packed once per engine from the module's parameters."""

    def __init__(self, text_model, device):
        from . import engine as E
        E._require_gpu(device)
        tm = getattr(text_model, "text_model", text_model)
        cfg = text_model.config
        if cfg.hidden_act != "quick_gelu":
            raise NotImplementedError(f"CLIP text tower with hidden_act={cfg.hidden_act!r}: only quick_gelu (ViT-L/14) is packed")
        self.device, self.c, self.heads = device, cfg.hidden_size, cfg.num_attention_heads
        self.dh, self.eps, self.ff = self.c // self.heads, float(cfg.layer_norm_eps), cfg.intermediate_size
        if self.c % 64 or self.ff % 64 or self.dh not in (32, 64, 128):
            raise NotImplementedError("CLIP text geometry outside what the folded-LayerNorm GEMM / md_attention cover")
        h = lambda t: t.detach().to(device=device, dtype=torch.float16).contiguous()   # noqa: E731
        f = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()   # noqa: E731
        emb = tm.embeddings
        self.tok, self.pos = h(emb.token_embedding.weight), h(emb.position_embedding.weight)
        self.layers = []
        G = 1.702
        for lyr in tm.encoder.layers:
            a = lyr.self_attn
            wqkv = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0)
            bqkv = torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0)
            self.layers.append(dict(
                qkv=E.fold_layernorm(wqkv, bqkv, lyr.layer_norm1.weight, lyr.layer_norm1.bias, device),
                o_w=h(a.out_proj.weight), o_b=f(a.out_proj.bias),
                fc1=E.fold_layernorm(lyr.mlp.fc1.weight.detach().float() * G, lyr.mlp.fc1.bias.detach().float() * G,
                                     lyr.layer_norm2.weight, lyr.layer_norm2.bias, device),
                fc2_w=h(lyr.mlp.fc2.weight.detach().float() / G), fc2_b=f(lyr.mlp.fc2.bias)))
        self.lnf = (f(tm.final_layer_norm.weight), f(tm.final_layer_norm.bias))
        self.ws = torch.zeros(32 << 20, dtype=torch.uint8, device=device)

    @torch.no_grad()
    def __call__(self, ids):
        from . import ops
        from .engine import NetEngine
        dev, c, H, dh, F16 = self.device, self.c, self.heads, self.dh, torch.float16
        ids = ids.to(dev)
        b, n = ids.shape
        if n > self.pos.shape[0]:
            raise ValueError(f"{n} tokens, the text tower has {self.pos.shape[0]} positions")
        new = lambda *shape: torch.empty(shape, dtype=F16, device=dev)   # noqa: E731
        x = new(b, n, c)
        ops.add_f16(self.tok.index_select(0, ids.reshape(-1)), self.pos[:n], x, b * n * c, b_period=n * c)
        x_lo = torch.zeros_like(x)
        ldv = (n + 7) & ~7
        gemm = dict(batch=b, hin=1, win=n, hout=1, wout=n, ws=self.ws)
        for L in self.layers:
            wl, s1, s0 = L["qkv"]
            qk, vt = new(b, n, 2 * c), torch.zeros(b, c, ldv, dtype=F16, device=dev)
            ops.igemm(x, wl, 3 * c, c0=c, ln=(s1, s0, self.eps), col_scale=(NetEngine.qscale(dh), c), out=qk, ld_out=2 * c,
                      out_t=vt, n_tr_begin=2 * c, ld_t=ldv, **gemm)
            att = new(b, n, c)
            ops.attention(qk, qk[:, :, c:], vt, att, batch=b, heads=H, nq=n, d=dh, n0=n, ld_q=2 * c, ld_k0=2 * c, ld_vt0=ldv,
                          ld_out=c, q_bs=n * 2 * c, k0_bs=n * 2 * c, vt0_bs=c * ldv, out_bs=n * c, q_prescaled=True, causal=True)
            y, y_lo = new(b, n, c), new(b, n, c)
            ops.igemm(att, L["o_w"], c, c0=c, bias=L["o_b"], res=x, ld_res=c, res_lo=x_lo, out=y, out_lo=y_lo, ld_out=c, **gemm)
            wl, s1, s0 = L["fc1"]
            hdn = new(b, n, self.ff)
            ops.igemm(y, wl, self.ff, c0=c, ln=(s1, s0, self.eps), act=ops.MD_ACT_SILU, out=hdn, ld_out=self.ff, **gemm)
            x, x_lo = new(b, n, c), new(b, n, c)
            ops.igemm(hdn, L["fc2_w"], c, c0=self.ff, bias=L["fc2_b"], res=y, ld_res=c, res_lo=y_lo, out=x, out_lo=x_lo, ld_out=c,
                      **gemm)
        out = new(b, n, c)
        ops.layernorm(x, self.lnf[0], self.lnf[1], out, b * n, c, self.eps)
        return out.float()


class FrozenCLIPEmbedder(nn.Module):
    LAYERS = ("last", "pooled", "hidden")

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, freeze=True, layer="last",
                 layer_idx=None, init_weights=True, text_config=None):
        super().__init__()
        from transformers import CLIPTextConfig, CLIPTextModel
        assert layer in self.LAYERS
        self.offline = False
        try:
            from transformers import CLIPTokenizer
            self.tokenizer = CLIPTokenizer.from_pretrained(version)
            # without a cache transformers 5 hands back an EMPTY tokenizer instead of raising: check it knows the CLIP vocabulary
            probe = self.tokenizer([""], padding="max_length", max_length=max_length, return_tensors="pt")["input_ids"][0]
            if len(self.tokenizer) < 49408 or int(probe[0]) != BOS_ID or int(probe[1]) != EOS_ID or int(probe[-1]) != EOS_ID:
                raise OSError("CLIP vocabulary not available")
        except Exception:  # noqa: BLE001 -- no cache / no network
            self.tokenizer, self.offline = EmptyPromptTokenizer(max_length), True
        if text_config is None and init_weights and not self.offline:
            try:
                self.transformer = CLIPTextModel.from_pretrained(version)
            except Exception:  # noqa: BLE001
                self.transformer = None
        else:
            self.transformer = None
        # where the text tower's weights come from: "pretrained" (HF cache), "checkpoint" (set by _load_from_state_dict once the
        # cond_stage_model.transformer.* keys of a model checkpoint have been loaded) or "random" -- an architecture-only module whose
        # output is meaningless; forward() refuses to run in that state unless the caller opted in (tests: allow_random_weights)
        self.weights_from = "pretrained" if self.transformer is not None else "random"
        self.allow_random_weights = text_config is not None
        if self.transformer is None:   # architecture only; weights arrive through load_state_dict (cond_stage_model.transformer.*)
            self.transformer = CLIPTextModel(CLIPTextConfig(**dict(VIT_L14_TEXT_CONFIG, **(text_config or {}))))
        self.device, self.max_length, self.layer, self.layer_idx = device, max_length, layer, layer_idx
        if layer == "hidden":
            assert layer_idx is not None and 0 <= abs(layer_idx) <= 12
        self._empty = None          # cached embedding of the empty prompt, [1, max_length, hidden]
        self._engine = None         # ClipTextEngine of the current weights (GPU only; rebuilt whenever the weights change; False:
        self._engine_fp = None      # this geometry is not served by the kernels -> transformers computes it)
        if freeze:
            self.freeze()

    def freeze(self):
        self.transformer = self.transformer.eval()
        for p in self.parameters():
            p.requires_grad = False

    def _apply(self, fn, *a, **k):   # .to() / .half() / .cuda(): the cached embedding follows the weights
        self._empty = self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._empty = self._engine = None
        return super().load_state_dict(*a, **k)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, *a, **k):
        self._empty = self._engine = None
        return super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, *a, **k)

    def note_loaded_keys(self, state_dict, prefix="cond_stage_model."):
        """called by the model's load_state_dict: the text tower counts as loaded when the checkpoint held every one of its
        tensors (a checkpoint without cond_stage_model.* keys, or with a key layout adapt_clip_keys did not map, leaves it random)"""
        want = set(prefix + k for k in self.state_dict())
        if want and want <= set(state_dict):
            self.weights_from = "checkpoint"

    @torch.no_grad()
    def _encode_ids(self, ids):
        dev = next(self.transformer.parameters()).device
        if dev.type == "cuda" and self.layer == "last" and self._engine is not False:   # the HIP kernels
            # the packed fp16 weights belong to ONE state of the parameters: (address, version counter) of every tensor -- an
            # in-place edit or a load_state_dict on the inner module (neither passes this wrapper's hooks) rebuilds the engine
            fp = tuple((p.data_ptr(), p._version) for p in self.transformer.parameters())
            if self._engine is None or self._engine_fp != fp:
                try:
                    self._engine, self._engine_fp = ClipTextEngine(self.transformer, dev), fp
                except NotImplementedError:
                    # a text tower this library's kernels do not serve (activation other than quick_gelu, head size not 32 / 64 / 128):
                    # the stock transformers module computes it -- the reference's own arithmetic for this (once-per-sequence) row
                    self._engine = False
            if self._engine is not False:
                return self._engine(ids)
        out = self.transformer(input_ids=ids.to(dev), output_hidden_states=self.layer == "hidden")
        if self.layer == "last":
            return out.last_hidden_state
        if self.layer == "pooled":
            return out.pooler_output[:, None, :]
        return out.hidden_states[self.layer_idx]

    @torch.no_grad()
    def forward(self, text):
        if getattr(self, "weights_from", None) == "random" and not getattr(self, "allow_random_weights", False):
            raise RuntimeError("FrozenCLIPEmbedder holds randomly initialised weights: neither a Hugging Face cache of "
                               "openai/clip-vit-large-patch14 nor cond_stage_model.transformer.* keys in the loaded checkpoint -- "
                               "pass --context_embedding (a saved [1, 77, 768] tensor) instead of a prompt")
        text = [text] if isinstance(text, str) else list(text)
        if all(t == "" for t in text):   # CLIP(""): computed once, repeated per sample
            if self._empty is None:
                enc = self.tokenizer([""], truncation=True, max_length=self.max_length, return_length=True,
                                     return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
                self._empty = self._encode_ids(enc["input_ids"])
            return self._empty.expand(len(text), -1, -1).clone()
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return self._encode_ids(enc["input_ids"])

    def encode(self, text):
        return self(text)
