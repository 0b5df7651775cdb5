# -*- coding: utf-8 -*-
'''
`with Message('...'):` indented block logging (reference: flow/message.py:12-24,
which wraps dolfin's begin()/end()).  Here a plain indent logger; silent unless
enabled with set_log_active(True).
'''
from __future__ import print_function

_STATE = {'level': 0, 'active': False}


def set_log_active(flag):
    _STATE['active'] = bool(flag)


def begin(string):
    info(string)
    _STATE['level'] += 1


def end():
    _STATE['level'] = max(0, _STATE['level'] - 1)


def info(string):
    if _STATE['active']:
        print('  ' * _STATE['level'] + str(string))


class Message(object):

    def __init__(self, string):
        self.string = string
        return

    def __enter__(self):
        begin(self.string)
        return

    def __exit__(self, tpe, value, traceback):
        end()
        return
