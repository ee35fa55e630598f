"""Stand-in for the reference's vendored unitrack package root (its real __init__ pulls torchvision/cv2)."""
