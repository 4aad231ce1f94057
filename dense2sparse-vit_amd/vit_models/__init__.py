"""Drop-in for the hot-path part of the reference's `vit_models` package (reference __init__.py:1-13 re-exports
every model file; only the dense-to-sparse path is provided here)."""
from .dynamic_vit import *  # noqa: F401,F403
from .dynamic_vit import (VisionTransformerDiffPruning, VisionTransformerTeacher, PredictorLG, Attention, Block, Mlp, PatchEmbed,  # noqa: F401
                          BatchNormLayer, batch_index_select, resize_pos_embed, checkpoint_filter_fn)
from .peturbed_topk import PerturbedTopK, PerturbedTopKFunction  # noqa: F401
