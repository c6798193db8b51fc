"""The same known answers evaluated through the HIP library (reference function names)."""
import pytest

import known_answers as ka

pytestmark = pytest.mark.gpu


def test_conversions(gpu):
    from fbx import operator_tools as ot
    ka.check_conversions(ot)


def test_projections(gpu):
    from fbx import operator_tools as ot
    ka.check_projections(ot)


def test_process_fidelity(gpu):
    from fbx import distance_measures as dm, operator_tools as ot
    ka.check_process_fidelity(ot, dm)
