"""ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

A minimal, build-owned stand-in for the third-party package ``ikpy==3.3.4``
(pinned by the reference at ``setup.py:14``; not installed in this image and
not installable -- no network).  It restates just the IKPy surface that
``seqikpy`` touches (``Chain``, ``OriginLink``, ``URDFLink``,
``forward_kinematics``, ``inverse_kinematics``) on top of the container's
real ``scipy.optimize.least_squares`` so that the reference's own, unmodified
``LegInvKinSeq`` / ``KinematicChainSeq`` sources can be imported from
``/root/reference`` and run here to generate golden vectors
(``oracle/gen_golden.py``).  Semantics: SURVEY.md Appendix A; validated
against the shipped ``leg_joint_angles.pkl`` (RF max 3.4e-5 rad).
"""
__version__ = "3.3.4-shim"
