function y = invAAt(x,p,q,sg1,sg2)
% Drop-in shim (invAAt.m:7-12 nargin rules resolved here); forwards to libipdamg.
if nargin == 3, sg1 = 1; sg2 = 1; end
if nargin == 4, sg2 = sg1; end
y = ipd_mex('invAAt', x, p, q, sg1, sg2);
end
