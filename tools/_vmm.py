"""HIP virtual-memory helpers for the placement probes (tools/place_split.py, place_pairs.py): physical chunks (hipMemCreate)
mapped back to back into one reserved address range and handed to torch through __cuda_array_interface__."""
import ctypes

hip = ctypes.CDLL('libamdhip64.so')


class Loc(ctypes.Structure):
    _fields_ = [('type', ctypes.c_int), ('id', ctypes.c_int)]


class Prop(ctypes.Structure):
    _fields_ = [('type', ctypes.c_int), ('handle', ctypes.c_int), ('location', Loc), ('win32', ctypes.c_void_p),
                ('compression', ctypes.c_ubyte), ('rdma', ctypes.c_ubyte), ('usage', ctypes.c_ushort)]


class Access(ctypes.Structure):
    _fields_ = [('location', Loc), ('flags', ctypes.c_int)]


def chk(rc, what):
    if rc != 0:
        raise RuntimeError('%s -> %d' % (what, rc))


PROP = Prop(1, 0, Loc(1, 0), None, 0, 0, 0)          # pinned device memory of device 0
_g = ctypes.c_size_t()
chk(hip.hipMemGetAllocationGranularity(ctypes.byref(_g), ctypes.byref(PROP), 0), 'hipMemGetAllocationGranularity')
GR = max(_g.value, 2 << 20)


def rnd(b):
    return (b + GR - 1) // GR * GR


def create(nbytes):
    h = ctypes.c_void_p()
    chk(hip.hipMemCreate(ctypes.byref(h), ctypes.c_size_t(nbytes), ctypes.byref(PROP), ctypes.c_ulonglong(0)), 'hipMemCreate(%d)' % nbytes)
    return h


class Mapped:
    """pieces: list of (handle, bytes), mapped back to back; unmapped (the chunks stay) when the object goes"""
    def __init__(self, shape, pieces):
        self.tot = sum(b for _, b in pieces)
        va = ctypes.c_void_p()
        chk(hip.hipMemAddressReserve(ctypes.byref(va), ctypes.c_size_t(self.tot), ctypes.c_size_t(0), ctypes.c_void_p(0), ctypes.c_ulonglong(0)), 'hipMemAddressReserve')
        off = 0
        for h, b in pieces:
            chk(hip.hipMemMap(ctypes.c_void_p(va.value + off), ctypes.c_size_t(b), ctypes.c_size_t(0), h, ctypes.c_ulonglong(0)), 'hipMemMap')
            off += b
        acc = Access(Loc(1, 0), 3)
        chk(hip.hipMemSetAccess(va, ctypes.c_size_t(self.tot), ctypes.byref(acc), ctypes.c_size_t(1)), 'hipMemSetAccess')
        self.ptr = va.value
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': '<f4', 'data': (self.ptr, False), 'version': 2}

    def close(self):
        if self.ptr:
            hip.hipMemUnmap(ctypes.c_void_p(self.ptr), ctypes.c_size_t(self.tot))
            hip.hipMemAddressFree(ctypes.c_void_p(self.ptr), ctypes.c_size_t(self.tot))
            self.ptr = 0
