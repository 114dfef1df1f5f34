"""`import mireg` -> the package that lives in the (un-importably named) directory
self-supervised-medical-image-registration-using-deep-optical-flow-estimation-with-brain-mri-data_amd/."""
import importlib
import os
import sys

PACKAGE_DIR = "self-supervised-medical-image-registration-using-deep-optical-flow-estimation-with-brain-mri-data_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module(PACKAGE_DIR)
sys.modules[__name__] = _pkg
