"""MI355X-native multi-model consensus inference engine.

Layers (bottom-up): ``ops`` (gfx950 HIP kernels + torch oracles) -> ``models`` ->
``engine`` (paged KV, continuous batching, hipGraph decode) -> ``consensus``
(protocol drivers and reducers) -> ``cli`` / ``serve``. ``parallel`` holds the RCCL
communication layer, tensor/expert parallel sharding and member placement.
"""
__version__ = "0.1.0"
