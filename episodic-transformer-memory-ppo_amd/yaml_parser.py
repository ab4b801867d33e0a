"""YAML config -> nested dict (same keys as upstream configs; PyYAML instead of ruamel which this image lacks)."""
import yaml


class YamlParser:
    def __init__(self, path):
        with open(path, "r") as stream:
            docs = [d for d in yaml.safe_load_all(stream) if d is not None]
        self._config = dict(docs[-1]) if docs else {}

    def get_config(self):
        return self._config
