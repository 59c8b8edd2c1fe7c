"""Exceptions raised by flow_amd (mirror of flow/utils/exceptions.py)."""


class FatalFlowError(Exception):
    """Unrecoverable error (reference: flow/utils/exceptions.py FatalFlowError)."""

    def __init__(self, msg='Unknown Fatal Error'):
        Exception.__init__(self, msg)
