"""MultiModalBertEncoder (ECAMP/Pre-training/module/bert_encoder.py:12-22)."""
import torch.nn as nn

from .bert_config import BertConfig
from .bert_modeling import MultimodalBertMaskedLM


class MultiModalBertEncoder(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        self.model = MultimodalBertMaskedLM(config if config is not None else BertConfig())

    def forward(self, latent, gap_token, ids, labels, attn_mask, token_type, weights, owner, B, T):
        return self.model(latent, gap_token, ids, attn_mask, token_type, weights, labels, owner, B, T)
