"""mi355 — host-side plumbing for the MI355X (gfx950) native library of the Carla-ppo hot path.

`lib`      ctypes binding generated from include/mi355_carla.h (fails loudly if the .so is missing).
`build`    hipcc --offload-arch=gfx950 build of csrc/*.hip into libmi355_carla.so (in-tree).
`dist`     one-process-per-GPU data parallelism over torch.distributed (RCCL on ROCm, gloo in CPU tests).
"""
