"""Loader for the golden vectors under tests/golden/ (generated from the upstream reference
by oracle/gen_golden.py)."""
import json
import os

from safetensors.torch import load_file

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self):
        with open(os.path.join(GOLDEN_DIR, "manifest.json")) as f:
            self.manifest = json.load(f)
        self._files = {}

    def tensors(self, name):
        if name not in self._files:
            self._files[name] = load_file(os.path.join(GOLDEN_DIR, name + ".safetensors"))
        return self._files[name]

    def cases(self, name):
        return self.manifest[name]["cases"]

    def case(self, name, key):
        """dict of the tensors of one case, with the `key.` prefix stripped"""
        t = self.tensors(name)
        pre = key + "."
        return {k[len(pre):]: v for k, v in t.items() if k.startswith(pre)}


def cases(name, section="cases"):
    with open(os.path.join(GOLDEN_DIR, "manifest.json")) as f:
        return json.load(f)[name][section]
