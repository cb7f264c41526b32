"""Text conditioning.  The CLIP ViT-L/14 text encoder runs ONCE per sequence (`get_learned_conditioning([""])`,
test_any_image_pose.py:196-198), far off the hot path, so its arithmetic stays stock `transformers` on PyTorch-ROCm as the survey
recommends; this module is the YAML ``cond_stage_config`` target with the reference's constructor kwargs, attribute names
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
        if freeze:
            self.freeze()

    def freeze(self):
        self.transformer = self.transformer.eval()
        for p in self.parameters():
            p.requires_grad = False

    def _apply(self, fn, *a, **k):   # .to() / .half() / .cuda(): the cached embedding follows the weights
        self._empty = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._empty = None
        return super().load_state_dict(*a, **k)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, *a, **k):
        self._empty = None
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
