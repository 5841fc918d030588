raise ImportError("the checkout's own lib/nets.py was imported: the dropin shadow lost the sys.path race")
