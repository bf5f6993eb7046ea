"""MultimodalBertModel / MultimodalBertMaskedLM (ECAMP/Pre-training/module/bert_modeling.py:9-227) on HIP stages."""
import types

import torch
import torch.nn as nn

from .bert_layers import BertEmbeddings, BertEncoder, BertOnlyMLMHead, BertPooler
from .context_fusion import ECAMPFusionLayer


class MultimodalBertModel(nn.Module):
    """embeddings -> context fusion with the image tokens -> 6 BertLayers (bert_modeling.py:113-131).
    Registration order follows the reference: embeddings, encoder, pooler, context_fusion_layer (:11-13)."""

    def __init__(self, config, add_pooling_layer=True):
        super().__init__()
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.pooler = BertPooler(config)  # the reference ignores add_pooling_layer (bert_modeling.py:11-12)
        self.context_fusion_layer = ECAMPFusionLayer(config)

    def forward(self, latent, gap_token, input_ids, attention_mask, token_type_ids, owner, B, T):
        from ..functions import BertEmbedFn, BertLayerFn
        cfg = self.config
        S = input_ids.shape[1]
        if S > cfg.max_position_embeddings:
            raise ValueError("sequence length %d exceeds max_position_embeddings %d" % (S, cfg.max_position_embeddings))
        pa = cfg.attention_probs_dropout_prob if self.training else 0.0
        ph = cfg.hidden_dropout_prob if self.training else 0.0
        key_mask = attention_mask.to(torch.int32).contiguous()  # additive finfo.min mask of bert_modeling.py:92, as a predicate
        e = BertEmbedFn.apply(input_ids, token_type_ids, self.embeddings, owner, ph, self.embeddings.LayerNorm.weight)
        h = self.context_fusion_layer(e, latent, gap_token, owner, B, S, T, key_mask)
        if getattr(owner, "keep_aux", False):   # parity checks: the fusion layer's and the encoder's outputs (bert_modeling.py:186-208)
            owner._aux_text = {"fused": h.detach().clone().view(B, S, -1)}
        for layer in self.encoder.layer:
            h = BertLayerFn.apply(h, layer, owner, B, S, key_mask, pa, ph)
        if getattr(owner, "keep_aux", False):
            owner._aux_text["seq_out"] = h.detach().clone().view(B, S, -1)
        return h


class MultimodalBertMaskedLM(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.bert = MultimodalBertModel(config, add_pooling_layer=False)
        self.cls = BertOnlyMLMHead(config)

    def forward(self, latent, gap_token, input_ids, attention_mask, token_type_ids, weights, labels, owner, B, T):
        from ..functions import MlmHeadFn
        seq = self.bert(latent, gap_token, input_ids, attention_mask, token_type_ids, owner, B, T)
        loss = MlmHeadFn.apply(seq, labels, weights, self.cls, owner)
        return types.SimpleNamespace(loss=loss[0], logits=None, hidden_states=None, attentions=None)
