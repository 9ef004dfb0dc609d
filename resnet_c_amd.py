"""Import shim: the package directory is ``resnet.c_amd/`` (the name the project
layout prescribes), which is not a legal dotted module name.  ``import
resnet_c_amd`` executes this file, which loads that directory as the package
``resnet_c_amd`` and replaces itself in ``sys.modules``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resnet.c_amd")
_spec = importlib.util.spec_from_file_location(
    "resnet_c_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["resnet_c_amd"] = _mod
_spec.loader.exec_module(_mod)
