"""Builds a JPEG that carries an EXIF block with an embedded thumbnail (IFD1 tags 256/257/513/514),
the layout cameras write and the reference's JPEG_EXIF_THUMBNAIL option reads (jpeg.inl:1654-1678, 4967-4976)."""
import struct


def with_exif_thumbnail(main_jpeg: bytes, thumb_jpeg: bytes, thumb_w: int, thumb_h: int, orientation: int = 6,
                        big_endian: bool = False, with_dims: bool = True) -> bytes:
    e = ">" if big_endian else "<"
    tiff = bytearray((b"MM" if big_endian else b"II") + struct.pack(e + "HI", 42, 8))
    # IFD0: one tag (orientation), then the offset of IFD1
    ifd0 = struct.pack(e + "H", 1) + struct.pack(e + "HHIHH", 274, 3, 1, orientation, 0)
    ifd1_off = 8 + len(ifd0) + 4
    tags = []
    if with_dims:
        tags += [(256, 4, 1, thumb_w), (257, 4, 1, thumb_h)]
    n1 = len(tags) + 2
    data_off = ifd1_off + 2 + 12 * n1 + 4
    tags += [(513, 4, 1, data_off), (514, 4, 1, len(thumb_jpeg))]
    ifd1 = struct.pack(e + "H", n1) + b"".join(struct.pack(e + "HHII", *t) for t in tags) + struct.pack(e + "I", 0)
    tiff += ifd0 + struct.pack(e + "I", ifd1_off) + ifd1 + thumb_jpeg
    app1 = b"Exif\x00\x00" + bytes(tiff)
    assert len(app1) + 2 < 65536
    assert main_jpeg[:2] == b"\xff\xd8"
    return b"\xff\xd8\xff\xe1" + struct.pack(">H", len(app1) + 2) + app1 + main_jpeg[2:]
