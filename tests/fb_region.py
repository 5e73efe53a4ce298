"""Framebuffer mode: how many bytes from the buffer's start the reference defines (the MCU rows it walks x the pitch of iCropCX pixels,
jpeg.inl:5114-5124); behind them it overruns the image and the product does not."""


def fb_defined_bytes(J, info, pixel_type, options, crop):
    bpp = {0: 2, 1: 2, 2: 4, 3: 1}[3 if (options & 64 and pixel_type < 3) else pixel_type]
    sh = 1 if options & 2 else 2 if options & 4 else 3 if options & 8 else 0
    mh = info.mcu_h >> sh
    cx, cy, cw, ch = J.crop_round(info, *crop) if crop else (0, 0, info.width, info.height)
    rows_mcu = min((cy + ch + info.mcu_h - 1) // info.mcu_h, info.mcus_y)
    kept = [y for y in range(rows_mcu) if y * mh >= cy]
    return (kept[-1] * mh - cy + mh) * cw * bpp if kept else 0
