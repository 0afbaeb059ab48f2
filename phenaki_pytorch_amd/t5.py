"""Text-encoder boundary (reference /root/reference/phenaki_pytorch/t5.py:64-103).

The T5 encoder itself is NOT part of the MI355X hot path: `Phenaki.sample` consumes a (B, L, d) f32 context whose
padded positions are zero-filled (the text mask is `any(embeds != 0)`, phenaki_pytorch.py:461).  This module keeps
the reference's `t5_encode_text` contract through HuggingFace transformers when weights are available, and the
`Phenaki.encode_texts` instance attribute stays the injection point for cached / precomputed embeddings
(`phenaki.encode_texts = lambda texts, output_device=None: cached`).
"""
import torch

DEFAULT_T5_NAME = 'google/t5-v1_1-base'
MAX_LENGTH = 256

_KNOWN_DIMS = {
    't5-small': 512, 't5-base': 768, 't5-large': 1024, 't5-3b': 1024, 't5-11b': 1024,
    'google/t5-v1_1-small': 512, 'google/t5-v1_1-base': 768, 'google/t5-v1_1-large': 1024,
    'google/t5-v1_1-xl': 2048, 'google/t5-v1_1-xxl': 4096,
}
_MODELS = {}


def get_encoded_dim(name):
    if name in _KNOWN_DIMS:
        return _KNOWN_DIMS[name]
    from transformers import T5Config
    return T5Config.from_pretrained(name).d_model


def _get(name):
    if name not in _MODELS:
        from transformers import T5EncoderModel, T5Tokenizer
        _MODELS[name] = (T5EncoderModel.from_pretrained(name).eval(), T5Tokenizer.from_pretrained(name))
    return _MODELS[name]


@torch.no_grad()
def t5_encode_text(texts, name=DEFAULT_T5_NAME, output_device=None):
    """List[str] -> (B, L, d) f32, pad positions zero-filled (t5.py:97-100)."""
    model, tok = _get(name)
    if torch.cuda.is_available():
        model = model.cuda()
    device = next(model.parameters()).device
    enc = tok.batch_encode_plus(texts, return_tensors='pt', padding='longest', max_length=MAX_LENGTH, truncation=True)
    input_ids, attn_mask = enc.input_ids.to(device), enc.attention_mask.to(device)
    out = model(input_ids=input_ids, attention_mask=attn_mask).last_hidden_state.detach()
    out = out.masked_fill(~attn_mask.bool()[..., None], 0.)
    return out if output_device is None else out.to(output_device)
