"""ECAMPFusionLayer -- parameter container with the reference's names and registration order
(ECAMP/Pre-training/module/context_fusion.py:8-19).  Its arithmetic (context_fusion.py:21-72) runs as ONE
hand-written stage, `ecamp_amd.functions.FusionFn`."""
import torch.nn as nn

from .bert_layers import BertAttention, BertIntermediate, BertOutput, BertSelfAttention, BertSelfOutput


class ECAMPFusionLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.cross_self_attention = BertSelfAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        self.gap_mlp = nn.Linear(config.hidden_size, config.hidden_size)
        self.out_layer = BertSelfOutput(config)

    def forward(self, hidden_states, encoder_hidden_states, gap_token, owner, B, S, T, key_mask):
        from ..functions import FusionFn
        cfg = owner.bert_config
        pa = cfg.attention_probs_dropout_prob if self.training else 0.0
        ph = cfg.hidden_dropout_prob if self.training else 0.0
        return FusionFn.apply(hidden_states, encoder_hidden_states, gap_token, self, owner, B, S, T, key_mask, pa, ph)
