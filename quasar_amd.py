"""Importable alias of the hyphen-named package directory."""
import importlib
import sys

_pkg = importlib.import_module("distributed-multi-agent-slam-swarm-robotics-system_amd")
sys.modules[__name__] = _pkg
