"""TensorBoard event files without tensorboard.

The reference logs scalars through torch.utils.tensorboard.SummaryWriter (DDQN.py:207,342-344;
ACKTR.py:185-188,401-421).  That package is not in this image, so this module writes the same
on-disk format itself: a TFRecord stream (`u64 length | masked crc32c(length) | payload | masked
crc32c(payload)`) of `Event` protobuf messages — first `file_version = "brain.Event:2"`, then one
`Event{wall_time, step, summary{value{tag, simple_value}}}` per scalar.  Only the handful of
fields scalars need are encoded (by hand: field numbers from tensorflow's event.proto / summary.proto).
`read_events` parses the same subset back and checks every checksum.  Parity with tensorboard's own
writer is unpinned (the package is absent); the format follows the published specification."""
import os
import socket
import struct
import time

_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data):
    """CRC-32C (Castagnoli), the checksum TFRecord uses."""
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field(num, wire, payload):
    return _varint((num << 3) | wire) + payload


def _len_delim(num, payload):
    return _field(num, 2, _varint(len(payload)) + payload)


def encode_scalar_event(tag, value, step, wall_time):
    value_msg = _len_delim(1, tag.encode("utf-8")) + _field(2, 5, struct.pack("<f", float(value)))   # Summary.Value
    summary = _len_delim(1, value_msg)                                                             # Summary
    return (_field(1, 1, struct.pack("<d", wall_time)) + _field(2, 0, _varint(int(step))) +         # Event
            _len_delim(5, summary))


def encode_version_event(wall_time):
    return _field(1, 1, struct.pack("<d", wall_time)) + _len_delim(3, b"brain.Event:2")


def frame(payload):
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", masked_crc(head)) + payload + struct.pack("<I", masked_crc(payload))


class EventFileWriter:
    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s.%d" % (time.time(), socket.gethostname(), os.getpid()))
        self._f = open(self.path, "ab")
        self._f.write(frame(encode_version_event(time.time())))
        self._f.flush()

    def add_scalar(self, tag, value, step, wall_time=None):
        self._f.write(frame(encode_scalar_event(tag, value, step, time.time() if wall_time is None else wall_time)))
        self._f.flush()

    def close(self):
        self._f.close()


# ---- reading back (tests, tools) ---------------------------------------------------------------
def _read_varint(buf, pos):
    n = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return n, pos


def _fields(buf):
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        num, wire = key >> 3, key & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wire == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v, pos = buf[pos:pos + n], pos + n
        else:
            raise ValueError("unsupported wire type %d" % wire)
        yield num, wire, v


def read_events(path):
    """[{wall_time, step, file_version | (tag, value)}] — raises on a bad checksum or framing."""
    out = []
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        head = data[pos:pos + 8]
        (n,) = struct.unpack("<Q", head)
        if struct.unpack("<I", data[pos + 8:pos + 12])[0] != masked_crc(head):
            raise ValueError("bad length checksum at %d" % pos)
        payload = data[pos + 12:pos + 12 + n]
        if struct.unpack("<I", data[pos + 12 + n:pos + 16 + n])[0] != masked_crc(payload):
            raise ValueError("bad payload checksum at %d" % pos)
        pos += 16 + n
        ev = {"step": 0}
        for num, wire, v in _fields(payload):
            if num == 1 and wire == 1:
                ev["wall_time"] = struct.unpack("<d", v)[0]
            elif num == 2 and wire == 0:
                ev["step"] = v
            elif num == 3 and wire == 2:
                ev["file_version"] = v.decode()
            elif num == 5 and wire == 2:
                for n1, w1, val in _fields(v):
                    if n1 == 1 and w1 == 2:
                        for n2, w2, x in _fields(val):
                            if n2 == 1 and w2 == 2:
                                ev["tag"] = x.decode()
                            elif n2 == 2 and w2 == 5:
                                ev["value"] = struct.unpack("<f", x)[0]
        out.append(ev)
    return out
