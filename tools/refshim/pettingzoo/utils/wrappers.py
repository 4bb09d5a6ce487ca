def CaptureStdoutWrapper(env):
    return env


def OrderEnforcingWrapper(env):
    return env
