"""The host-side post-processing parity tests (NMS behind the C ABI, the evaluator, the array-shaped results) once more
under the `gpu` marker, so that the GPU box's `pytest -m gpu` run exercises them with the library it built -- they need no
device, and the CPU suite runs the originals (tests/test_postproc_eval.py)."""
import pytest

from test_postproc_eval import (test_batched_nms_equals_per_query_oracle, test_evaluator_array_path_matches_reference,  # noqa: F401
                                test_evaluator_matches_reference, test_fuzz_nms_vs_oracle,
                                test_moment_results_roundtrip_and_protocol, test_nms_edge_cases, test_nms_matches_reference)

pytestmark = pytest.mark.gpu
