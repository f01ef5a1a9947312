"""`Unet` of the reference's models/dam/model_unet_MandD4.py (ablation of the direction-aware-mask head: mask + 4+1-class direction;
forward :246-268).  Same constructor, state_dict keys and return tuple; encoder, decoder and residual units are the kernels of
model_unet_rev1, the heads are plain 1x1 classifiers (cdnet_final_conv1x1).  Trains through
cdnet_amd.trainer.AblationTrainer (cdnet_dam_loss_classes with direction_classes = 5)."""
from .model_unet_MandD import Unet as _MandD


class Unet(_MandD):
    DIRECTION_OUT = 5
