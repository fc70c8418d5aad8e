"""Binary PLY reader / writer with the interface and the on-disk layout of the reference's
KPConv-PyTorch/utils/ply.py (read_ply :116-196, write_ply :217-328): one packed little/big-endian record per
vertex, header property types spelled as NumPy dtype names (float32, uint8, int32, ...), optional
`property list uchar int vertex_indices` triangle block. Files written by either implementation are read
by the other; for the same arrays the bytes are identical (tests/test_host_cpu.py, fixtures written by the
reference itself: tests/golden/g9_*.ply)."""
import sys

import numpy as np

# header type word -> NumPy type code (both the PLY spellings and the NumPy names occur in the wild)
_TYPE_CODES = {
    'int8': 'i1', 'char': 'i1', 'uint8': 'u1', 'uchar': 'u1',
    'int16': 'i2', 'short': 'i2', 'uint16': 'u2', 'ushort': 'u2',
    'int32': 'i4', 'int': 'i4', 'uint32': 'u4', 'uint': 'u4',
    'float32': 'f4', 'float': 'f4', 'float64': 'f8', 'double': 'f8',
}
_BYTE_ORDER = {'binary_little_endian': '<', 'binary_big_endian': '>'}


def _read_header(stream):
    """-> (byte order prefix, {element name: count}, vertex property list [(name, dtype str)])."""
    if b'ply' not in stream.readline():
        raise ValueError('The file does not start whith the word ply')
    fmt = stream.readline().split()[1].decode()
    if fmt == 'ascii':
        raise ValueError('The file is not binary')
    order = _BYTE_ORDER[fmt]
    counts, vertex_props, element = {}, [], None
    while True:
        line = stream.readline()
        if line == b'' or b'end_header' in line:
            break
        words = line.split()
        if not words:
            continue
        if words[0] == b'element':
            element = words[1].decode()
            counts[element] = int(words[2])
        elif words[0] == b'property':
            if words[1] == b'list':
                if element == 'vertex':
                    raise ValueError('Unsupported vertex list property : ' + line.decode().strip())
                continue
            if element in (None, 'vertex'):
                vertex_props.append((words[2].decode(), order + _TYPE_CODES[words[1].decode()]))
    return order, counts, vertex_props


def read_ply(filename, triangular_mesh=False):
    """Structured array with one field per vertex property; with triangular_mesh=True the list
    [vertex data, faces (n,3) int32]."""
    with open(filename, 'rb') as stream:
        order, counts, props = _read_header(stream)
        n_vertex = counts.get('vertex', next(iter(counts.values())) if counts else 0)
        vertices = np.fromfile(stream, dtype=props, count=n_vertex)
        if not triangular_mesh:
            return vertices
        face_record = [('k', order + 'u1'), ('v1', order + 'i4'), ('v2', order + 'i4'), ('v3', order + 'i4')]
        raw = np.fromfile(stream, dtype=face_record, count=counts.get('face', 0))
        return [vertices, np.stack([raw['v1'], raw['v2'], raw['v3']], axis=1)]


def write_ply(filename, field_list, field_names, triangular_faces=None):
    """Every 1-D array and every column of a 2-D array in field_list is one vertex property, named by
    field_names in order. Returns True, or False (after printing why) when the fields do not fit --
    the reference's behaviour."""
    fields = list(field_list) if isinstance(field_list, (list, tuple)) else [field_list]
    columns = []
    for arr in fields:
        arr = np.asarray(arr)
        if arr.ndim > 2:
            print('fields have more than 2 dimensions')
            return False
        arr = arr.reshape(-1, 1) if arr.ndim < 2 else arr
        columns.extend(arr[:, j] for j in range(arr.shape[1]))
    if len({c.shape[0] for c in columns}) > 1:
        print('wrong field dimensions')
        return False
    if len(columns) != len(field_names):
        print('wrong number of field names')
        return False
    if not filename.endswith('.ply'):
        filename += '.ply'
    n = columns[0].shape[0] if columns else 0
    header = ['ply', 'format binary_%s_endian 1.0' % sys.byteorder, 'element vertex %d' % n]
    header += ['property %s %s' % (c.dtype.name, name) for c, name in zip(columns, field_names)]
    if triangular_faces is not None:
        header += ['element face %d' % triangular_faces.shape[0], 'property list uchar int vertex_indices']
    header.append('end_header')
    record = np.empty(n, dtype=[(name, c.dtype.str) for c, name in zip(columns, field_names)])
    for c, name in zip(columns, field_names):
        record[name] = c
    with open(filename, 'wb') as stream:
        stream.write(('\n'.join(header) + '\n').encode('ascii'))
        record.tofile(stream)
        if triangular_faces is not None:
            tri = np.asarray(triangular_faces).astype(np.int32)
            faces = np.empty(tri.shape[0], dtype=[('k', 'uint8'), ('0', 'int32'), ('1', 'int32'), ('2', 'int32')])
            faces['k'] = 3
            faces['0'], faces['1'], faces['2'] = tri[:, 0], tri[:, 1], tri[:, 2]
            faces.tofile(stream)
    return True
