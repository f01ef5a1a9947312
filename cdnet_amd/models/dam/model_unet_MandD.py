"""`Unet` of the reference's models/dam/model_unet_MandD.py (ablation of the direction-aware-mask head: mask + 9-class direction, no attention gates, no point branch;
forward :246-268).  Same constructor, state_dict keys and return tuple; encoder, decoder and residual units are the kernels of
model_unet_rev1, the heads are plain 1x1 classifiers (cdnet_final_conv1x1).  Training: cdnet_amd.trainer.AblationTrainer."""
from .model_unet_rev1 import Unet as _Rev1


class Unet(_Rev1):
    VARIANT = 'MandD'
    DIRECTION_OUT = 9
    # parameters the reference's forward never touches (no gradient, never stepped)
    UNUSED_PREFIXES = ('final_conv.', 'child0.', 'child_conv1.', 'point_feature.', 'point_conv.', 'directionAtt.', 'maskAtt.')
