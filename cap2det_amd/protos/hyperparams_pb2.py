"""Stand-in for the protoc-generated `protos/hyperparams_pb2.py` of the reference (schema.py)."""
from cap2det_amd.protos.schema import (  # noqa: F401
    Hyperparams, Regularizer, L1Regularizer, L2Regularizer, Initializer, TruncatedNormalInitializer, VarianceScalingInitializer, RandomNormalInitializer, GlorotNormalInitializer, GlorotUniformInitializer, BatchNorm)
