"""Import-time stand-in for `pygame` (rendering is out of scope)."""


def __getattr__(name):
    if name.startswith("K_"):
        return hash(name) & 0xFFFF
    raise AttributeError(name)


def init():
    pass
