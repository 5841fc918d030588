"""Stands for the checkout's own lib/utils.py: a module the MI355X package does NOT shadow -- `from lib import utils` must
still find it (dropin/lib/__init__.py appends the checkout's lib/ to its __path__)."""
ORIGIN = 'checkout'
