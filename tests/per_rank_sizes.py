"""The per-rank batches `bench.py --gpus N` produces for the workloads that shard a FIXED global size (c4: 256 homotopy levels of
1 024 segments; c5 / c5_stm: 65 536 segments), N = 1, 2, 4, 8, and the STM kernel family LTO_KERNEL_AUTO resolves to at each on an
MI355X (256 CUs, default cost table).  Shared by tests/test_auto_kernel.py (CPU: the pure function lto_indirect_auto_kernel must give
exactly these, and bench.py's sharding these sizes) and the GPU parity tests that run every one of them against the oracle
(tests/test_gpu_baseline_shapes.py), so that the 8-GPU box, which the builder cannot touch, never meets a family at a size class the
one-GPU suite has not checked."""
WORLDS = (1, 2, 4, 8)
# workload -> {world: (segments per rank, family of the STM sweep)}
C4 = {1: (262144, "segment-lane"), 2: (131072, "segment-lane"), 4: (65536, "segment-lane"), 8: (32768, "pipeline48")}
C5_STM = {1: (65536, "cooperative2"), 2: (32768, "cooperative2"), 4: (16384, "cooperative2"), 8: (8192, "cooperative2")}
C2 = {w: (4096, "pipeline8") for w in WORLDS}          # weak scaling: every rank its own 4 096 segments, 14-dim
