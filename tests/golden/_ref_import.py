"""Import the read-only reference (PengNi/ccsmeth @ /root/reference) in THIS container only.

Used only by the fixture generators in tests/golden/ (never by tests, bench or the product).
The reference imports pysam/tabix/pybedtools/statsmodels at module scope
(ccsmeth/utils/process_utils.py:7, ccsmeth/extract_features.py), none of which are installed here
and none of which the model/feature arithmetic needs, so empty stand-in modules are registered.
"""
import os
import sys
import types

REF_ROOT = "/root/reference"


def import_reference():
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (fixture generation only works in the build container)")
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    for name in ("pysam", "tabix", "pybedtools"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if "statsmodels" not in sys.modules:
        sm = types.ModuleType("statsmodels")
        sm.robust = types.ModuleType("statsmodels.robust")
        sys.modules["statsmodels"] = sm
        sys.modules["statsmodels.robust"] = sm.robust
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import ccsmeth  # noqa: F401
    return ccsmeth
