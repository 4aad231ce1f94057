"""Drop-in for the hot-path part of the reference's `vit_models` package (reference __init__.py:1-13 re-exports
every model file; only the dense-to-sparse path and the T2T path are provided here)."""
from .dynamic_vit import *  # noqa: F401,F403
from .dynamic_vit import (VisionTransformerDiffPruning, VisionTransformerTeacher, PredictorLG, Attention, Block, Mlp, PatchEmbed,  # noqa: F401
                          BatchNormLayer, batch_index_select, resize_pos_embed, checkpoint_filter_fn)
from .peturbed_topk import PerturbedTopK, PerturbedTopKFunction  # noqa: F401
from .token_performer import Token_performer  # noqa: F401
from .token_transformer import Token_transformer  # noqa: F401
from .transformer_block import get_sinusoid_encoding  # noqa: F401
from .t2t_vit import (T2T_ViT, T2T_module, T2t_vit_14, T2t_vit_t_14, T2T_ViT_DiffPruning, T2T_ViT_Teacher,  # noqa: F401
                      t2t_vit_14_student, t2t_vit_14_teacher)
