"""Experiment-manifest loading without TensorFlow and without importing user modules.

The reference opens `os.path.join(text_dir, experiment_manifest_name)` with
`yaml.full_load` (ecog2txt/trainers.py:60-61).  Its shipped manifests carry python tags
(`!!python/name:...DataGenerator`, `!!python/tuple`, `!!set`; e.g.
auxiliary/EFC/mocha-1_word_sequence.yaml:4,8,34) that full_load can only resolve by importing
the named module -- which needs TF or a private lab module.  Here the tags are resolved
against a registry of DataGenerator classes instead (falling back to the TF-free
ECoGDataGenerator shell), so the reference's own manifests load unchanged."""
import importlib
import os
import sys

import yaml

_REGISTRY = {}
_ALLOW_IMPORT = False


def register_data_generator(dotted_name, cls):
    """Make `!!python/name:<dotted_name>` resolve to `cls`."""
    _REGISTRY[dotted_name] = cls


class UnresolvedName:
    """Placeholder for a python name that could not be imported (kept so the manifest still loads)."""

    def __init__(self, dotted):
        self.dotted = dotted

    def __repr__(self):
        return 'UnresolvedName(%r)' % self.dotted


def _resolve(dotted):
    if dotted in _REGISTRY:
        return _REGISTRY[dotted]
    tail = dotted.rsplit('.', 1)[-1]
    # the reference's own shell class and its usual subclasses map onto this package's classes
    from . import data_generators as dg
    if hasattr(dg, tail):
        return getattr(dg, tail)
    # like yaml.full_load (the reference, trainers.py:60-61): only names of modules that are ALREADY imported; loading a
    # manifest never imports -- i.e. executes -- a module (allow_import=True in load_manifest opts in)
    mod, _, name = dotted.rpartition('.')
    m = sys.modules.get(mod)
    if m is None and _ALLOW_IMPORT:
        try:
            m = importlib.import_module(mod)
        except Exception:
            m = None
    if m is not None and hasattr(m, name):
        return getattr(m, name)
    return UnresolvedName(dotted)


class ManifestLoader(yaml.SafeLoader):
    pass


def _name_constructor(loader, suffix, node):
    return _resolve(suffix)


def _tuple_constructor(loader, node):
    return tuple(loader.construct_sequence(node))


ManifestLoader.add_multi_constructor('tag:yaml.org,2002:python/name:', _name_constructor)
ManifestLoader.add_constructor('tag:yaml.org,2002:python/tuple', _tuple_constructor)


def load_manifest(path_or_name, text_dir=None, allow_import=False):
    """Load an experiment manifest: {subject_id: {key: value}}.  allow_import: let a
    `!!python/name:` tag import the module it names (off by default: a manifest is data)."""
    global _ALLOW_IMPORT
    path = path_or_name
    if text_dir is not None and not os.path.isabs(path_or_name):
        path = os.path.join(text_dir, path_or_name)
    prev, _ALLOW_IMPORT = _ALLOW_IMPORT, bool(allow_import)
    try:
        with open(path) as f:
            return yaml.load(f, Loader=ManifestLoader)
    finally:
        _ALLOW_IMPORT = prev
