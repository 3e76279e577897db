"""Resource lookup with the reference's search order
(pisa/utils/resources.py:42-60): absolute / CWD-relative path, then each
directory of the colon-separated ``PISA_RESOURCES`` environment variable, then
the resources packaged with this build (``pisa_amd/resources``)."""
import os

PACKAGED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "resources")


def find_resource(resource, fail=True):
    resource = os.path.expandvars(os.path.expanduser(str(resource)))
    if os.path.isabs(resource) or os.path.exists(resource):
        if os.path.exists(resource):
            return resource
    roots = [p for p in os.environ.get("PISA_RESOURCES", "").split(":") if p] + [PACKAGED]
    for root in roots:
        for sub in ("", "data", "scripts", "settings"):
            cand = os.path.join(root, sub, resource)
            if os.path.exists(cand):
                return cand
    if fail:
        raise IOError('Could not find resource "%s"' % resource)
    return None
