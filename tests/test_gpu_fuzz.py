"""A slice of the randomised differential campaign (tools/fuzz_parity.py): random scenes, cameras, builders and options, the HIP
path tracer and the hybrid passes against the CPU oracle, bit for bit.  The long runs are in profiles/r02_fuzz_parity.json
(18 k cases, no findings); this keeps 120 fixed seeds in the driver's GPU pass."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.gpu
def test_random_scenes_match_oracle():
    import fuzz_parity

    findings = []
    rays = 0
    for seed in range(424242, 424242 + 120):
        info, problems = fuzz_parity.run_case(seed)
        rays += info.get("rays", 0)
        if problems:
            findings.append((info, problems))
    assert not findings, findings[:3]
    assert rays > 100000  # the cases do trace something
