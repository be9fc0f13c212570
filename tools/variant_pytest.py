"""pytest against another build of the library: python tools/variant_pytest.py <lib.so> [pytest arguments...] (measurement / experiment variants:
tools/_variants; the product suite always loads minppo_amd/libminppo_hip.so)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat
nat.HIP_LIB_PATH = Path(sys.argv[1]).resolve()
import pytest
raise SystemExit(pytest.main(sys.argv[2:]))
