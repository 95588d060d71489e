"""MI355X-native EfficientDet forward/backward path (hand-written HIP for gfx950 behind the
reference's nn.Module surface).  See DESIGN.md / INTEGRATION.md."""
