"""BertConfig with the reference's defaults (ECAMP/Pre-training/module/bert_config.py:63-94).  Plain Python:
the product does not depend on `transformers` (the reference pins 4.42.4; 5.x already broke its imports)."""


class BertConfig:
    model_type = "bert"

    def __init__(self, vocab_size=30000, hidden_size=768, num_hidden_layers=6, num_attention_heads=6, intermediate_size=1536,
                 hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=256,
                 type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12, pad_token_id=0,
                 position_embedding_type="absolute", use_cache=True, classifier_dropout=None, **kwargs):
        if hidden_act != "gelu":
            raise ValueError("only the exact-erf 'gelu' activation of the reference config is implemented")
        if position_embedding_type != "absolute":
            raise ValueError("only absolute position embeddings are implemented")
        if hidden_size % num_attention_heads != 0 or hidden_size // num_attention_heads not in (32, 64, 128):
            raise ValueError("head_dim must be 32, 64 or 128")
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.hidden_act = hidden_act
        self.intermediate_size = intermediate_size
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.initializer_range = initializer_range
        self.layer_norm_eps = layer_norm_eps
        self.pad_token_id = pad_token_id
        self.position_embedding_type = position_embedding_type
        self.use_cache = use_cache
        self.classifier_dropout = classifier_dropout
        for k, v in kwargs.items():
            setattr(self, k, v)
