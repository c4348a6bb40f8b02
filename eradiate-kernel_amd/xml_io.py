"""Mitsuba 2 scene XML -> scene dictionary.

The reference parses XML in C++ (/root/reference/src/libcore/xml.cpp); this module turns the same documents into the
dictionaries `load_dict` already understands, so `load_string` / `load_file` build scenes through one code path.
Covered: object tags with `type` / `id` / `name`, the property tags (float, integer, boolean, string, point, vector,
rgb, spectrum with a single value, transform with translate / rotate / scale / lookat / matrix), `<ref>`, `<default>` +
`$parameter` substitution (xml.cpp:259-284), `<include>` (xml.cpp:524-561), unnamed children numbered `_arg_N` in
document order (xml.cpp:619-620).  Not covered: version upgrades of 0.x scenes, `<alias>`, `<path>`, wavelength-value
spectra (they need the spectral upsampling the rgb path does not have).
"""
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

from .transform import ScalarTransform4f

OBJECT_TAGS = {"scene", "sensor", "film", "rfilter", "sampler", "integrator", "emitter", "shape", "bsdf", "medium", "phase",
               "volume", "texture"}
PROPERTY_TAGS = {"float", "integer", "boolean", "string", "point", "vector", "rgb", "spectrum", "transform", "ref", "default",
                 "include", "translate", "rotate", "scale", "lookat", "matrix"}
MAX_INCLUDE_RECURSION = 15          # MTS_XML_INCLUDE_MAX_RECURSION, include/mitsuba/core/xml.h:8
RESERVED_IDS = re.compile(r"^_unnamed_\d+$")


class XMLError(RuntimeError):
    pass


def _err(src, msg):
    raise XMLError('Error while loading "%s": %s' % (src, msg))


def _subst(value, params, src):
    """$name substitution in attribute values (xml.cpp:259-284); longest names first."""
    if "$" not in value:
        return value
    for k in sorted(params, key=len, reverse=True):
        value = value.replace("$" + k, str(params[k]))
    if "$" in value:
        _err(src, 'undefined default parameter in "%s"' % value)
    return value


def _floats(s, src, what, n=None):
    toks = [t for t in re.split(r"[\s,]+", s.strip()) if t]
    try:
        vals = [float(t) for t in toks]
    except ValueError:
        _err(src, 'could not parse floating point value "%s"' % s)
    if n is not None and len(vals) != n:
        _err(src, '%s: expected %d values, got "%s"' % (what, n, s))
    return vals


def _vec3(node, attrs, src, default=None):
    """x / y / z attributes or value="a, b, c" (a single value is broadcast), xml.cpp:304-340."""
    if "value" in attrs:
        if any(k in attrs for k in "xyz"):
            _err(src, 'can\'t mix and match "value" and "x"/"y"/"z" attributes')
        v = _floats(attrs["value"], src, node.tag)
        if len(v) == 1:
            v = v * 3
        if len(v) != 3:
            _err(src, '"value" attribute must have exactly 1 or 3 elements')
        return v
    out = []
    for k in "xyz":
        if k in attrs:
            out.append(_floats(attrs[k], src, node.tag, 1)[0])
        elif default is not None:
            out.append(default)
        else:
            _err(src, 'missing attribute "%s" in "%s"' % (k, node.tag))
    return out


def _check_attrs(node, attrs, allowed, required, src):
    for k in attrs:
        if k not in allowed:
            _err(src, 'unexpected attribute "%s" in "%s"' % (k, node.tag))
    for k in required:
        if k not in attrs:
            _err(src, 'missing attribute "%s" in "%s"' % (k, node.tag))


def _transform(node, params, src):
    tr = ScalarTransform4f()
    for child in node:
        a = {k: _subst(v, params, src) for k, v in child.attrib.items()}
        if child.tag == "translate":
            op = ScalarTransform4f.translate(_vec3(child, a, src, 0.0))
        elif child.tag == "scale":
            op = ScalarTransform4f.scale(_vec3(child, a, src, 1.0))
        elif child.tag == "rotate":
            _check_attrs(child, a, {"x", "y", "z", "angle", "value"}, {"angle"}, src)
            axis = _vec3(child, {k: v for k, v in a.items() if k != "angle"}, src, 0.0)
            op = ScalarTransform4f.rotate(axis, _floats(a["angle"], src, "rotate", 1)[0])
        elif child.tag == "lookat":
            _check_attrs(child, a, {"origin", "target", "up"}, {"origin", "target"}, src)
            origin, target = _floats(a["origin"], src, "lookat", 3), _floats(a["target"], src, "lookat", 3)
            if "up" in a:
                up = _floats(a["up"], src, "lookat", 3)
            else:                                                        # xml.cpp:843-849: any vector perpendicular to the view direction
                from .transform import coordinate_system
                d = np.asarray(target, np.float32) - np.asarray(origin, np.float32)
                up = coordinate_system((d / np.linalg.norm(d)).astype(np.float32))[0]
            op = ScalarTransform4f.look_at(origin, target, up)
        elif child.tag == "matrix":
            _check_attrs(child, a, {"value"}, {"value"}, src)
            v = _floats(a["value"], src, "matrix")
            if len(v) == 9:
                m = np.eye(4, dtype=np.float32)
                m[:3, :3] = np.asarray(v, np.float32).reshape(3, 3)
            elif len(v) == 16:
                m = np.asarray(v, np.float32).reshape(4, 4)
            else:
                _err(src, "matrix: expected 16 or 9 values")
            op = ScalarTransform4f(m)
        else:
            _err(src, 'unexpected element "%s" inside a transform' % child.tag)
        tr = op @ tr                                                     # xml.cpp:815: later operations act on the left
    return tr


class _Parser:
    def __init__(self, src, params, base_dir, include_depth=0):
        self.src, self.params, self.base_dir = src, dict(params), base_dir
        self.ids = {}
        self.include_depth = include_depth                               # XMLSource::depth, xml.cpp:662-673

    def obj(self, node, depth=0):
        a = {k: _subst(v, self.params, self.src) for k, v in node.attrib.items()}
        if node.tag == "scene":
            _check_attrs(node, a, {"version", "id", "name"}, set(), self.src)
            out = {"type": "scene"}
        else:
            _check_attrs(node, a, {"type", "id", "name"}, {"type"}, self.src)
            out = {"type": a["type"]}
        if "id" in a:
            if RESERVED_IDS.match(a["id"]):
                _err(self.src, 'invalid id "%s" in "%s": ids starting with "_unnamed_" are reserved' % (a["id"], node.tag))
            if a["id"] in self.ids:
                _err(self.src, '"%s" has duplicate id "%s"' % (node.tag, a["id"]))
            self.ids[a["id"]] = node.tag
            out["id"] = a["id"]
        arg = 0
        for child in node:
            ca = {k: _subst(v, self.params, self.src) for k, v in child.attrib.items()}
            tag = child.tag
            if tag == "default":
                _check_attrs(child, ca, {"name", "value"}, {"name", "value"}, self.src)
                if depth != 0:
                    _err(self.src, '"default" elements are only allowed at the root of the scene')
                self.params.setdefault(ca["name"], ca["value"])
                continue
            if tag == "path":                                            # xml.cpp:633-650: a search path for the FileResolver
                _check_attrs(child, ca, {"value"}, {"value"}, self.src)
                if depth != 0:
                    _err(self.src, "<path>: path can only be child of root")
                from .fresolver import file_resolver
                cand = ca["value"] if os.path.isabs(ca["value"]) else os.path.join(self.base_dir, ca["value"])
                if not os.path.exists(cand):
                    cand = file_resolver().resolve(ca["value"])
                if not os.path.exists(cand):
                    _err(self.src, '<path>: folder "%s" not found' % cand)
                file_resolver().prepend(cand)
                continue
            if tag == "include":
                _check_attrs(child, ca, {"filename"}, {"filename"}, self.src)
                path = ca["filename"] if os.path.isabs(ca["filename"]) else os.path.join(self.base_dir, ca["filename"])
                if not os.path.exists(path):
                    _err(self.src, 'included file "%s" not found' % ca["filename"])
                if self.include_depth + 1 > MAX_INCLUDE_RECURSION:       # xml.cpp:671-673
                    raise XMLError("Exceeded <include> recursion limit of %d" % MAX_INCLUDE_RECURSION)
                sub = _Parser(path, self.params, os.path.dirname(path), self.include_depth + 1)
                sub.ids = self.ids
                try:
                    root = ET.parse(path).getroot()
                except (ET.ParseError, OSError) as exc:                  # xml.cpp:675-678
                    _err(self.src, 'error while loading "%s": %s' % (path, exc))
                inc = sub.obj(root, depth)
                items = inc.items() if root.tag == "scene" else [("_arg_%d" % arg, inc)]
                for k, v in items:
                    if k in ("type",):
                        continue
                    if k.startswith("_arg_"):
                        k = "_arg_%d" % arg
                        arg += 1
                    out[k] = v
                continue
            if tag in OBJECT_TAGS or tag == "ref":
                if tag == "ref":
                    _check_attrs(child, ca, {"id", "name"}, {"id"}, self.src)
                    value = {"type": "ref", "id": ca["id"]}
                else:
                    value = self.obj(child, depth + 1)
                name = ca.get("name")
                if name is None:
                    # The reference finds unnamed children by their class (e.g. Sensor::Sensor looks for a Film among
                    # props.objects()); the dictionary loader finds them by key, so a lone film / sampler / rfilter / bsdf /
                    # phase / emitter / medium child takes its tag as its name.  Everything else keeps the `_arg_N` of xml.cpp.
                    unique = tag in ("film", "sampler", "rfilter", "bsdf", "phase", "emitter", "medium") and node.tag != "scene" \
                        and out.get("type") != "blendphase" and tag not in out and sum(1 for c in node if c.tag == tag and "name" not in c.attrib) == 1
                    if unique:
                        name = tag
                    else:
                        name = "_arg_%d" % arg
                        arg += 1
            elif tag in PROPERTY_TAGS:
                if "name" not in ca:
                    _err(self.src, 'missing attribute "name" in "%s"' % tag)
                name = ca["name"]
                value = self.prop(child, ca)
            else:
                _err(self.src, 'unexpected element "%s"' % tag)
            if name.startswith("_") and not name.startswith("_arg_"):
                _err(self.src, 'invalid parameter name "%s" in "%s": leading underscores are reserved' % (name, tag))
            if name in out:
                _err(self.src, 'Property "%s" was specified multiple times!' % name)
            out[name] = value
        return out

    def prop(self, node, a):
        tag = node.tag
        if tag in ("float", "integer", "boolean", "string"):
            _check_attrs(node, a, {"name", "value"}, {"value"}, self.src)
            v = a["value"].strip()
            if tag == "string":
                return a["value"]
            if tag == "boolean":
                if v.lower() not in ("true", "false"):
                    _err(self.src, 'could not parse boolean value "%s" -- must be "true" or "false"' % v)
                return v.lower() == "true"
            if tag == "integer":
                if not re.fullmatch(r"[+-]?\d+", v):
                    _err(self.src, 'could not parse integer value "%s"' % v)
                return int(v)
            if not re.fullmatch(r"[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?|inf|nan)", v):
                _err(self.src, 'could not parse floating point value "%s"' % v)
            return float(v)
        if tag in ("point", "vector"):
            _check_attrs(node, a, {"name", "value", "x", "y", "z"}, set(), self.src)
            return _vec3(node, {k: v for k, v in a.items() if k != "name"}, self.src)
        if tag == "rgb":
            _check_attrs(node, a, {"name", "value"}, {"value"}, self.src)
            v = _floats(a["value"], self.src, "rgb")
            if len(v) == 1:
                v = v * 3
            if len(v) != 3:
                _err(self.src, "'rgb' tag requires one or three values (got \"%s\")" % a["value"])
            return {"type": "rgb", "value": v}
        if tag == "spectrum":
            _check_attrs(node, a, {"name", "value", "filename"}, set(), self.src)
            if ("value" in a) == ("filename" in a):              # xml.cpp:815-816
                _err(self.src, "'spectrum' tag requires one of \"value\" or \"filename\" attributes")
            if "filename" in a:                                  # spectrum_from_file (libcore/spectrum.cpp:9-39), read by the loader through the FileResolver
                return {"type": "spectrum", "filename": a["filename"]}
            if ":" in a["value"]:                                # wavelength:value pairs (xml.cpp:560-600): regular / irregular in the spectral variant
                return {"type": "spectrum", "value": a["value"]}
            return {"type": "spectrum", "value": _floats(a["value"], self.src, "spectrum", 1)[0]}
        if tag == "transform":
            _check_attrs(node, a, {"name"}, set(), self.src)
            return _transform(node, self.params, self.src)
        _err(self.src, 'unexpected element "%s"' % tag)


def xml_to_dict(string, params=None, src="<string>", base_dir="."):
    try:
        root = ET.fromstring(string)
    except ET.ParseError as e:
        _err(src, "XML parse error: %s" % e)
    if root.tag not in OBJECT_TAGS:
        if root.tag in PROPERTY_TAGS:
            _err(src, 'root element "%s" must be an object' % root.tag)
        _err(src, 'unexpected root element "%s"' % root.tag)
    parser = _Parser(src, params or {}, base_dir)
    d = parser.obj(root)
    _check_refs(d, parser.ids, src)
    return d


def _check_refs(d, ids, src):
    for v in d.values():
        if isinstance(v, dict):
            if v.get("type") == "ref":
                if v["id"] not in ids:
                    _err(src, 'reference to unknown object "%s"!' % v["id"])
            else:
                _check_refs(v, ids, src)


def file_to_dict(path, params=None):
    if not os.path.exists(path):
        raise XMLError('"%s": file does not exist!' % path)
    with open(path, "r") as fh:
        return xml_to_dict(fh.read(), params, src=path, base_dir=os.path.dirname(os.path.abspath(path)))
