# -*- coding: utf-8 -*-
'''Which values that enter the graph keys move from step to step on the
plateau?  Prints the fields of the matrix-free Jacobian's struct, the bytes of
the preconditioner structs and the workspace addresses per step.'''
import ctypes
import os
import sys
import zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), 'tests'))


def main():
    from flow_amd import _hip
    from flow_amd.fem import ops
    import test_graph_replay as T
    prob = T._problem()
    _hip.graph_mode(1)
    prob.settle()
    lib = _hip.lib()
    seen = []
    orig = lib.flow_gmres_solve

    def crc(obj):
        return '%08x' % zlib.crc32(ctypes.string_at(ctypes.addressof(obj),
                                                    ctypes.sizeof(obj)))
    lay = prob.W.layout
    for k in range(6):
        s0 = _hip.graph_stats()
        prob.step()
        s1 = _hip.graph_stats()
        row = {}
        for key, val in lay._dev.items():
            if isinstance(key, tuple) and key and key[0] == 'jvp_operator':
                st = val.struct
                row['jvp'] = dict(mesh=ctypes.addressof(st.mesh.contents),
                                  W=ctypes.addressof(st.W.contents),
                                  bfmask=st.bfmask, ui=st.ui, scratch=st.scratch,
                                  nbc=st.nbc, bc_dofs=st.bc_dofs, bc_mask=st.bc_mask,
                                  mesh_crc=crc(st.mesh.contents),
                                  W_crc=crc(st.W.contents),
                                  op=crc(val._op))
        for slot in ('jacobian_pmg', 'jacobian_ilu'):
            pre = lay._dev.get(slot)
            if pre is not None and hasattr(pre, 'struct'):
                row[slot] = crc(pre.struct)
        print('step %d: captures cg %d gmres %d mass %d' % (
            k, s1['captures_cg'] - s0['captures_cg'],
            s1['captures_gmres'] - s0['captures_gmres'],
            s1['captures_mass'] - s0['captures_mass']))
        for name, val in row.items():
            print('    ', name, val)


if __name__ == '__main__':
    main()
