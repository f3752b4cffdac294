"""factorizer_amd — MI355X-native drop-in for the hot path of pashtari/factorizer.

``import factorizer_amd as ft`` exposes the reference's flat names for that path
(factorizer/__init__.py:1-6): ft.NMF, ft.SWMatricize, ft.FactorizerBlock, ft.Factorizer, ...
Device tensors run hand-written gfx950 kernels through libfactorizer_hip.so
(include/factorizer_hip.h); there is no silent fallback when the library is missing.
"""
from .utils import as_tuple, has_args, is_partializable, partialize
from .matricize import Matricize, Reshape, SWMatricize
from .nmf import (NMF, SVD, BCDSolver, Compose, CoordinateDescent, FastMultiplicativeUpdate, Initializer,
                  LeastSquares, MatrixFactorization, MultiplicativeUpdate, NNDSVDInit, ProjectedGradient,
                  RandomInit, SemiMultiplicativeUpdate, SVDInit, WeightedMultiplicativeUpdate, relative_error)
from .layers import MLP, LayerNorm, Linear, PosEmbed, PositionalEmbedding
from .convs import Conv3d, ConvTranspose3d
from .blocks import FactMixer, FactorizerBlock, FactorizerStage
from .losses import DiceCELoss, dice_bce_loss, dice_ce_loss
from .training import FlatAdamW, WarmupCosineSchedule, load_checkpoint, load_checkpoints
from .parallel import FlatGradSync
from .inference import SlidingWindowInferer, SlidingWindowInfererAdapt, sliding_window_inference
from .ushape import (Factorizer, Same, UNet, UNetDecoder, UNetDecoderBlock, UNetEncoder,
                   UNetEncoderBlock, UNetStage)
from .deconver import Deconv, DeconvInitializer, DeconvMixer, Deconver, DeconverBlock, DeconverStage, Stem

__version__ = "0.1.0"
