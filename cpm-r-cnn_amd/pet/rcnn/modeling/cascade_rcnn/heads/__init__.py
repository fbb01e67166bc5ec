from .mlp_heads import roi_2mlp_head
