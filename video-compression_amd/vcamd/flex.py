"""placeholder -- replaced below in this round"""
