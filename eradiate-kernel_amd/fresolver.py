"""FileResolver (include/mitsuba/core/fresolver.h, src/libcore/fresolver.cpp): an ordered list of search paths through which
plugins resolve relative file names (`ply` / `obj`: src/shapes/ply.cpp:107-108, `gridvolume`: src/textures/volume_data.h:48-49).
The reference keeps one per thread (Thread::file_resolver()); this host mirror keeps one per process."""
import os


class FileResolver:
    def __init__(self, paths=None):
        self._paths = list(paths) if paths is not None else [os.getcwd()]

    def __len__(self):
        return len(self._paths)

    def __iter__(self):
        return iter(self._paths)

    def clear(self):
        self._paths = []

    def append(self, path):
        self._paths.append(str(path))

    def prepend(self, path):
        self._paths.insert(0, str(path))

    def resolve(self, path):
        """fresolver.cpp: an absolute path is returned as it is; otherwise the first search path that holds it wins; a name that
        cannot be found is returned unchanged (the plugin then reports the missing file)."""
        path = str(path)
        if not os.path.isabs(path):
            for base in self._paths:
                cand = os.path.join(base, path)
                if os.path.exists(cand):
                    return cand
        return path


_resolver = FileResolver()


def file_resolver():
    return _resolver


def set_file_resolver(r):
    global _resolver
    _resolver = r
