"""Regenerates the ctypes `ConvDesc` snippet of INTEGRATION.md from srvp_amd/_lib.py (the binding the tests hold against
include/srvp_hip.h), so the documented struct cannot fall behind the header.  `python tools/gen_integration_snippet.py`
rewrites the block between the BEGIN/END markers; tests/test_host.py::test_integration_snippet_is_current checks it."""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BEGIN, END = '<!-- BEGIN ConvDesc (generated: tools/gen_integration_snippet.py) -->', '<!-- END ConvDesc -->'


def _ctype_name(t):
    if hasattr(t, '_length_') and hasattr(t, '_type_'):
        return f'{_ctype_name(t._type_)} * {t._length_}'
    return {C.c_void_p: 'C.c_void_p', C.c_int32: 'C.c_int32', C.c_int64: 'C.c_int64', C.c_float: 'C.c_float',
            C.c_double: 'C.c_double'}[t]


def snippet():
    from srvp_amd import _lib as L
    items = [f"('{n}', {_ctype_name(t)})" for n, t in L.ConvDesc._fields_]
    lines, cur = [], '    _fields_ = ['
    for it in items:
        if len(cur) + len(it) + 2 > 124:
            lines.append(cur.rstrip())
            cur = '                '
        cur += it + ', '
    lines.append(cur.rstrip(', ') + ']')
    body = '\n'.join(lines)
    return (f"{BEGIN}\n```python\nimport ctypes as C, torch\nlib = C.CDLL('srvp_amd/libsrvp_hip.so')\n"
            f"class ConvDesc(C.Structure):                      # mirrors srvp_conv_desc field by field (tests check sizeof/offsets)\n"
            f"{body}\n"
            "lib.srvp_conv_mfma.argtypes = [C.POINTER(ConvDesc), C.c_void_p]; lib.srvp_conv_mfma.restype = C.c_int\n"
            "d = ConvDesc(); d.src0 = act.data_ptr(); ...      # bf16 NHWC tensors with a 1-pixel zero border; unset fields are 0\n"
            "rc = lib.srvp_conv_mfma(C.byref(d), torch.cuda.current_stream().cuda_stream)\n"
            "if rc: raise RuntimeError(lib.srvp_last_error().decode())\n```\n" + END)


def current(text):
    m = re.search(re.escape(BEGIN) + r'.*?' + re.escape(END), text, re.S)
    return m.group(0) if m else None


if __name__ == '__main__':
    p = os.path.join(ROOT, 'INTEGRATION.md')
    text = open(p).read()
    new = snippet()
    if current(text) is None:
        sys.exit('markers not found in INTEGRATION.md')
    open(p, 'w').write(text.replace(current(text), new))
    print('INTEGRATION.md ConvDesc snippet regenerated')
