"""--opt NAME=VALUE for the tools: per-context tuning options of the library (ccqp_ctx_set_option; tools/README.md).
The library reads no environment variable; a tool that wants an A/B switch says so on its command line and records it."""


def add_option_args(ap):
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="tuning option of the engine (repeatable), e.g. --opt MPRGP_LOOKAHEAD=1 --opt BMV3_SHAPE=444")


def apply_options(args, device=None):
    """Sets the options on the default engine of `device` and on every engine created afterwards; returns {name: value}."""
    from ccqppy_amd._engine import set_default_option
    out = {}
    for item in args.opt:
        name, _, value = item.partition("=")
        set_default_option(name.strip().upper(), int(value, 0), device)
        out[name.strip().upper()] = int(value, 0)
    return out
