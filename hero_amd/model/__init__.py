"""Drop-in module tree mirroring the reference's `model/` package (same class, attribute and
state-dict names) with forwards on the libhero_hip.so kernels."""
from .encoder import (CrossModalTrm, QueryFeatEncoder, RobertaModelConfig,  # noqa: F401
                      RobertaPreTrainedModel, TemporalTrm)
from .model import HeroModel, HierarchicalVlModel, VideoModelConfig, VideoPreTrainedModel  # noqa: F401
from .pretrain import HeroForPretraining  # noqa: F401
from .vcmr import HeroForVcmr  # noqa: F401
