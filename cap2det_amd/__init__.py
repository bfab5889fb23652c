"""cap2det_amd — MI355X-native hot path of Cap2Det (WSOD training step).

Only the hot path of SURVEY.md §8 lives here: HIP kernels + C-ABI (`csrc/`, `include/`)
and the host-side mirror of the reference's plugin interface (`models/`, `core/`,
`protos/`, `train/` keep the reference's module names so call sites read the same).
"""
__version__ = "0.1.0"
