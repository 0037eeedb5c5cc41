"""MI355X-native posterior-compute path for the acoustic models of OrcusCZ/NNAcousticModeling.

Python host over a C-ABI shared library (``libamfwd.so``, ``include/amfwd.h``) of hand-written HIP kernels for gfx950.
Importing the package does not load the library; the first use of ``engine.Engine`` does, and fails loudly when the
library or the GPU is missing -- there is no CPU fallback.
"""

__all__ = ["engine", "chainer_npz", "kaldi_ft", "kw_io", "networks", "predict_folds", "evaluate", "evaluate_model_for_test", "sharding"]
__version__ = "0.1"
