"""Measurement code behind `bench.py` (imported by it and by the tests; never by the product path):

    secondary      the single-GPU entries beside the headline line (L = 80, fp32 storage, API path, alignment, grounding, the training step)
    sharded_step   `bench.py --workload train_step [--gpus N]`: the training step of vlgae_amd/train_step.py sharded data-parallel,
                   bucketed RCCL all-reduce of every trainable parameter's gradient
"""
