"""A fixed slice of tools/stress_paths.py in the suite: the first 120 cases of seed 20261004 - generated scenes of random size,
seed and flags (translucent, textured, closed), random profiles (odd sizes, 1-9 spp, 0-7 bounces, every tone map), random
shards, tile shapes and sample batches - rendered by the default pipeline (origin grids, camera-grid cull, split shade pass,
hand-over kernel), the KD-tree pipeline and the megakernel: the same bits on all three, and the small frames among them equal
to the CPU oracle's.  (The tool itself runs as long as it is given; its first run found a hit the KD walk lost)"""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_random_scenes_profiles_and_shards_agree_on_three_paths_and_with_the_oracle():
    sys.path.insert(0, str(ROOT / "tools"))
    import stress_paths
    cases, culled, with_oracle, _ = stress_paths.run(300.0, 20261004, max_cases=120, verbose=False)
    assert 120 <= cases < 124 and culled > 60 and with_oracle >= 4, (cases, culled, with_oracle)
