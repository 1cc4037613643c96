"""Experiment directory object with the interface the reference gets from `mag.experiment.Experiment`
(train_2d_cnn.py:194-232 creates one from a config dict, :276-277 registers sub-directories, :365 registers
results; predict_2d_cnn.py:66 re-opens one with `resume_from`; networks/classifiers.py:803,822,846-866 read
`.summaries` / `.checkpoints`).  Directory layout (reference README.md:136-146):

    experiments/<identifier>/{config.json, results.json, command, commit_hash, log, checkpoints/, predictions/, summaries/}

`mag` itself (requirements.txt:100, a git dependency pinned to no version) is absent from this image and from
/root/reference, so the identifier rule is restated from its published behaviour -- parameters whose name starts with
an underscore are left out, the rest are flattened, abbreviated to the initials of their underscore-separated words
and joined as `<value><sep><abbreviation>` with "|" between them, sorted by abbreviation -- and is NOT pinned by any
fixture.  Nothing on the hot path depends on the identifier text: an existing directory (written by `mag` or by
this class) is opened by path (`resume_from`), and every file name inside it is the reference's.
"""
import json
import os
import subprocess
import sys

SEPARATOR = ["="]


def use_custom_separator(sep):
    """mag.use_custom_separator (train_2d_cnn.py:30 passes "-")."""
    SEPARATOR[0] = sep


class Config(dict):
    """Nested attribute-style view of a config dictionary (`config.network.conv_base_depth`)."""

    def __init__(self, mapping=()):
        super().__init__()
        for k, v in dict(mapping).items():
            self[k] = Config(v) if isinstance(v, dict) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def to_dict(self):
        return {k: v.to_dict() if isinstance(v, Config) else v for k, v in self.items()}

    def flat(self, prefix=""):
        out = {}
        for k, v in self.items():
            key = prefix + k
            if isinstance(v, Config):
                out.update(v.flat(key + "."))
            else:
                out[key] = v
        return out

    @property
    def identifier(self):
        parts = []
        for key, value in self.flat().items():
            names = key.split(".")
            if any(n.startswith("_") for n in names):
                continue
            abbrev = "".join(word[0] for word in names[-1].split("_") if word)
            parts.append((abbrev, "%s%s%s" % (value, SEPARATOR[0], abbrev)))
        return "|".join(text for _, text in sorted(parts))


def _set_dotted(tree, key, value):
    *path, leaf = key.split(".")
    for name in path:
        tree = tree.setdefault(name, {})
    tree[leaf] = value


class _Tee:
    def __init__(self, stream, path):
        self.stream, self.file = stream, open(path, "a")

    def write(self, text):
        self.stream.write(text)
        self.file.write(text)

    def flush(self):
        self.stream.flush()
        self.file.flush()

    def isatty(self):
        return False


class Experiment:
    """`with Experiment(config_dict, implicit_resuming=...) as experiment:` or `Experiment(resume_from=path)`."""

    def __init__(self, config=None, resume_from=None, implicit_resuming=False, experiments_dir="experiments",
                 write=True):
        if (config is None) == (resume_from is None):
            raise ValueError("pass either a config dictionary or resume_from")
        self._write = write
        if resume_from is not None:
            self.directory = resume_from
            with open(os.path.join(resume_from, "config.json")) as f:
                self.config = Config(json.load(f))
        else:
            self.config = Config(config)
            self.directory = os.path.join(experiments_dir, self.config.identifier)
            if os.path.isdir(self.directory) and os.path.isfile(os.path.join(self.directory, "config.json")):
                if not implicit_resuming:
                    raise ValueError("experiment %r already exists (pass --resume to continue it)" % self.directory)
            elif write:
                os.makedirs(self.directory, exist_ok=True)
                with open(os.path.join(self.directory, "config.json"), "w") as f:
                    json.dump(self.config.to_dict(), f, indent=4)
                with open(os.path.join(self.directory, "command"), "w") as f:
                    f.write(" ".join(sys.argv) + "\n")
                with open(os.path.join(self.directory, "commit_hash"), "w") as f:
                    f.write(self._commit_hash() + "\n")
        self._results = {}
        path = os.path.join(self.directory, "results.json")
        if os.path.isfile(path):
            with open(path) as f:
                self._results = json.load(f)
        for name in ("checkpoints", "predictions", "summaries"):
            if os.path.isdir(os.path.join(self.directory, name)):
                setattr(self, name, os.path.join(self.directory, name))
        self._tee = None

    @staticmethod
    def _commit_hash():
        try:
            return subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip()
        except (OSError, subprocess.SubprocessError):
            return ""

    def __enter__(self):
        if self._write:
            self._tee = (sys.stdout, _Tee(sys.stdout, os.path.join(self.directory, "log")))
            sys.stdout = self._tee[1]
        return self

    def __exit__(self, *exc):
        if self._tee is not None:
            sys.stdout = self._tee[0]
            self._tee[1].file.close()
            self._tee = None
        return False

    def register_directory(self, name):
        path = os.path.join(self.directory, name)
        if self._write:
            os.makedirs(path, exist_ok=True)
        setattr(self, name, path)

    def register_result(self, key, value):
        """`fold0.metric` nests under `fold0` (train_2d_cnn.py:457 later tests `"fold0" in results.to_dict()`)."""
        _set_dotted(self._results, key, float(value) if hasattr(value, "__float__") else value)
        if self._write:
            with open(os.path.join(self.directory, "results.json"), "w") as f:
                json.dump(self._results, f, indent=4)

    @property
    def results(self):
        return Config(self._results)
