"""Stand-in for the protoc-generated `protos/cap2det_model_pb2.py` of the reference (schema.py)."""
from cap2det_amd.protos.schema import (  # noqa: F401
    Cap2DetModel, TextModel)
