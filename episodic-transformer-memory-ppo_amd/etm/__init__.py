"""Host-side bindings of libetm_hip.so (hand-written gfx950 kernels) -- see include/etm_hip.h."""
