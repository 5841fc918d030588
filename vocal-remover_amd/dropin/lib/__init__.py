"""Shadow of the reference's `lib` package: put vocal-remover_amd/dropin on sys.path ahead of the
reference checkout and `from lib import nets, spec_utils, dataset` resolves to the MI355X path."""
