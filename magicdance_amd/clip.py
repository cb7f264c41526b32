"""Text conditioning: the CLIP ViT-L/14 text encoder runs ONCE per sequence (`get_learned_conditioning([""])`),
far off the hot path, so it stays stock `transformers` on PyTorch-ROCm as the survey recommends -- this module only removes
the dependency on the reference tree for the YAML ``cond_stage_config`` target.  Same constructor kwargs, attribute names
(``tokenizer`` / ``transformer`` -> checkpoint keys ``cond_stage_model.transformer.*``) and ``encode`` / ``forward`` behaviour
as model_lib/ControlNet/ldm/modules/encoders/modules.py:88-131.

Tokenizer vocabulary and (unless a checkpoint provides them) weights come from the Hugging Face cache of ``version``;
when they are absent (as in the build image: no network, no cache) construction raises and the model carries an
``_Unavailable`` placeholder -- callers then pass the [B,77,768] context tensor directly (``--context_embedding``).
``init_weights=False`` builds the architecture from its config only (weights to be filled by ``load_state_dict``)."""
import torch
import torch.nn as nn


class FrozenCLIPEmbedder(nn.Module):
    LAYERS = ("last", "pooled", "hidden")

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, freeze=True, layer="last",
                 layer_idx=None, init_weights=True):
        super().__init__()
        from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
        assert layer in self.LAYERS
        self.tokenizer = CLIPTokenizer.from_pretrained(version)
        if init_weights:
            self.transformer = CLIPTextModel.from_pretrained(version)
        else:
            self.transformer = CLIPTextModel(CLIPTextConfig.from_pretrained(version))
        self.device, self.max_length, self.layer, self.layer_idx = device, max_length, layer, layer_idx
        if layer == "hidden":
            assert layer_idx is not None and 0 <= abs(layer_idx) <= 12
        if freeze:
            self.freeze()

    def freeze(self):
        self.transformer = self.transformer.eval()
        for p in self.parameters():
            p.requires_grad = False

    @torch.no_grad()
    def forward(self, text):
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        dev = next(self.transformer.parameters()).device
        out = self.transformer(input_ids=enc["input_ids"].to(dev), output_hidden_states=self.layer == "hidden")
        if self.layer == "last":
            return out.last_hidden_state
        if self.layer == "pooled":
            return out.pooler_output[:, None, :]
        return out.hidden_states[self.layer_idx]

    def encode(self, text):
        return self(text)
