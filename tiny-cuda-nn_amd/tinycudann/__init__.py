"""tinycudann -- MI355X (gfx950) native drop-in for the HashGrid + FullyFusedMLP hot path of NVlabs/tiny-cuda-nn.

`import tinycudann as tcnn` gives the reference's PyTorch surface (`tcnn.NetworkWithInputEncoding`,
`tcnn.Network`, `tcnn.Encoding`, ...; reference bindings/torch/tinycudann/__init__.py) backed by the
hand-written HIP kernels in ../csrc through the C ABI of include/tcnn_hip.h.  `tcnn.native` adds the
C++ API's `create_from_config / trainer.training_step / network.inference` view.
"""
from . import _C  # noqa: F401  (raises ImportError if libtcnn_hip.so is missing)
from .modules import Encoding, Module, Network, NetworkWithInputEncoding, free_temporary_memory, rtc_set_cache_dir, supports_jit_fusion  # noqa: F401
from . import native  # noqa: F401
from .native import create_from_config  # noqa: F401
